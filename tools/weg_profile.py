"""Developer tool: N evaluations of the WEG objective + gradient at the product shape (B=1, L=16), for
`rocprofv3 --kernel-trace --stats -- python3 tools/weg_profile.py`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import weg  # noqa: E402

dev = torch.device("cuda:0")
model = bench.make_model(dev)
g = torch.Generator().manual_seed(9)
S = (24, 161, 24, 8, 1)
enc = [torch.randn(1, s, 512, generator=g).to(dev) for s in S]
masks = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].to(dev)}
lat = torch.randn(1, 16, 128, generator=g).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
weg.loss_and_grad(model, lat, 500, enc, masks, [[3, 9, 14]])
torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    weg.loss_and_grad(model, lat, 500, enc, masks, [[3, 9, 14]])
torch.cuda.synchronize()
print(f"{(time.time() - t0) / n * 1e3:.2f} ms per evaluation")
