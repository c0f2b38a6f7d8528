// bf16 matrix-core (MFMA) shift-GEMM - placeholder until the tiled kernel lands in this file.
#include "itts_kernels.h"
namespace itts {
bool gemm_mfma_supported(const GemmArgs&, int, int, int) { return false; }
int gemm_mfma(const GemmArgs&, int, int, int, hipStream_t) {
  set_error("gemm_mfma: unsupported shape");
  return E_INVALID;
}
}  // namespace itts
