#!/bin/bash
# Timing ablations of xattn_role_kernel (developer tool, GPU box; the ablated builds compute garbage): XR_ABLATE bit 1 = no
# fragment reads / MFMAs / softmax, 2 = no fills, 4 = no L2 touches.  Build: hipcc ... -DXR_ABLATE=<v> -o tools/experiments/lib_xr<v>.so
cd "$(dirname "$0")/../../.."
# (round-3 tree only: xattn_role_kernel and the CFD_XA_ROLE knob left the product sources in round 4; on a later tree this measures the stock kernel)
grep -q xattn_role convofusion_amd/csrc/cfd_api.hip || { echo "this tree has no xattn_role_kernel: check out the round-3 tree (tools/experiments/r03_variants/README)"; exit 1; }
for v in base role 4 1 5 2 6; do
  unset CFD_LIB; export CFD_XA_ROLE=1
  if [ $v = base ]; then export CFD_XA_ROLE=0; elif [ $v != role ]; then export CFD_LIB=$PWD/tools/experiments/lib_xr$v.so; fi
  python tools/xa_ablate.py 2>/dev/null | tail -1 | sed "s/^/variant $v: /"
done
