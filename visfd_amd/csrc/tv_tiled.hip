// tv_tiled.hip -- tiled tensor-voting kernel (placeholder until the LDS-tiled version lands).
#include "common.hpp"

namespace vh {

int dev_tv_tiled(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                 const float* mask_src, const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0,
                 i64 z_out1, int h, const float* w, const float* rhat, int exponent, bool curves,
                 bool* handled) {
  (void)ctx; (void)sal; (void)dir; (void)ten; (void)mask_src; (void)mask_dst; (void)nx; (void)ny;
  (void)nz; (void)z_out0; (void)z_out1; (void)h; (void)w; (void)rhat; (void)exponent; (void)curves;
  *handled = false;
  return VISFD_HIP_OK;
}

}  // namespace vh
