// libcfdenoise: developer / test hooks (include/cfdenoise_dev.h) -- stage-wise taps and internal buffers (the GEMM test / micro-benchmark
// hooks live in cfd_forward.hip, next to the product instances they launch).
#include "cfd_internal.hpp"

// ---- test hooks -----------------------------------------------------------------------------------------
extern "C" int cfd_debug_stop_stage(cfd_handle c, int stage) {
  if (!c) return fail(CFD_E_ARG, "null handle");
  c->stop_stage = stage;
  return CFD_OK;
}

extern "C" int cfd_debug_read(cfd_handle c, const char* what, float* dst_dev, size_t numel) {
  if (!c || !what || !dst_dev) return fail(CFD_E_ARG, "null argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!strcmp(what, "setup_launches")) {   // launches the last cfd_sample_begin spent on timestep-only tables (0: all served from the cache)
    const float f = (float)c->setup_launches;
    HIPCHK(hipMemcpy(dst_dev, &f, 4, hipMemcpyHostToDevice));
    return CFD_OK;
  }
  if (!strcmp(what, "sat")) {   // the saturation census as one float (not cleared)
    unsigned int n[2] = {0, 0};
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(n, c->sat.p, 8, hipMemcpyDeviceToHost));
    const float f = (float)n[0] + (float)n[1];
    HIPCHK(hipMemcpy(dst_dev, &f, 4, hipMemcpyHostToDevice));
    return CFD_OK;
  }
#if RT_STAMP
  if (!strcmp(what, "rt_ring")) {   // developer build: the launch time line (rowtile.hpp), 4 x 4096 64-bit words + the sequence counter
    HIPCHK(hipDeviceSynchronize());
    if (numel * 4 < sizeof(unsigned long long) * 4 * 4096 + 8) return fail(CFD_E_ARG, "rt_ring needs %zu bytes", sizeof(unsigned long long) * 4 * 4096 + 8);
    HIPCHK(hipMemcpyFromSymbol(dst_dev, HIP_SYMBOL(g_rt_ring), sizeof(unsigned long long) * 4 * 4096, 0, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpyFromSymbol(reinterpret_cast<char*>(dst_dev) + sizeof(unsigned long long) * 4 * 4096, HIP_SYMBOL(g_rt_seq), 4, 0, hipMemcpyDeviceToDevice));
    return CFD_OK;
  }
#endif
  const DBuf* b = nullptr;
  if (!strcmp(what, "x")) b = &c->w->x;
  else if (!strcmp(what, "temb")) b = &c->w->temb_tab;
  else if (!strcmp(what, "ss")) b = &c->w->ss_tab;
  else if (!strcmp(what, "eps")) b = &c->w->eps;
  else if (!strcmp(what, "sc")) b = &c->w->sc;
  else if (!strcmp(what, "ssc")) b = &c->w->ssc;
  else if (!strcmp(what, "xa_stamps")) b = &c->w->xa_stamps;
  else return fail(CFD_E_ARG, "unknown buffer '%s'", what);
  if (numel * 4 > b->bytes) return fail(CFD_E_ARG, "buffer '%s' holds %zu bytes, asked for %zu", what, b->bytes, numel * 4);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(dst_dev, b->p, numel * 4, hipMemcpyDeviceToDevice));
  return CFD_OK;
}
