// K9c, fp16 instantiations: the implicit-GEMM tile kernel of igemm_kernel.h on IEEE fp16 rows (HIAST_FMT_FP16) — the
// reference's apex-O1 arithmetic (code/utils/default_config.py:109, utils/utils.py:126-132: half-precision convolutions,
// fp32 accumulation, fp32 master weights) for the teacher forward, the student forward and the data gradients.
// Same kernel text as the bf16 variants (LDS-DMA slabs, asm fragment reads, counted waits); only H16<true> differs.
#include "igemm_kernel.h"

int hiast_igemm_launch_f16(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N, int taps,
                           hiast::IGeo geo, float* stats, const void* res_gate, int gate_mask, hipStream_t st, int stats_mode,
                           int out_f32, int stats_rows)
{
    if (out_f32)
        return launch_igemm_t<1, true, true>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats,
                                             res_gate, gate_mask, st, stats_mode, stats_rows);
    return launch_igemm_t<1, false, true>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats,
                                          res_gate, gate_mask, st, stats_mode, stats_rows);
}
