// rt_conv_gemm, K-parity ping-pong LDS-DMA variants (PIPE = 2 of gemm_dma_body, rt_gemm_dma.h; hints 3xx): two 4-wave groups per
// workgroup alternate between a memory segment (fragment reads + operand DMA + counted waits) and a pure MFMA segment, one raw
// barrier per K tile, four LDS stages, three tiles in flight.  Own translation unit so that it compiles beside rt_gemm.hip.
#include "rt_gemm_dma.h"

#ifdef RT_LAB        // measured slower inside the step than the pipelined forms (profiles/r06_pingpong_gemm.txt): lab library only
int rt_launch_gemm_pp(const GemmArgs& a, int hint, hipStream_t s) {
    if ((a.N & 7) || !a.epi_lds) return RT_ERR_UNSUPPORTED;       // the groups' partial sums meet in the LDS-staged epilogue
    switch (hint) {
        // (tile, stages, min workgroups / CU, waves, form)
        case 351: return launch_gemm_dma<128, 128, 4, 1, 8, 2>(a, s);
        case 321: return launch_gemm_dma<128, 64, 4, 1, 8, 2>(a, s);
        case 323: return launch_gemm_dma<64, 128, 4, 1, 8, 2>(a, s);
        case 331: return launch_gemm_dma<64, 64, 4, 2, 8, 2>(a, s);
        // three stages (96 KB at 128 x 128: leaves room for another stream's workgroup beside it), two tiles in flight
        case 352: return launch_gemm_dma<128, 128, 3, 1, 8, 2>(a, s);
        case 322: return launch_gemm_dma<128, 64, 3, 2, 8, 2>(a, s);
        case 332: return launch_gemm_dma<64, 64, 3, 3, 8, 2>(a, s);
        default: return RT_ERR_BADARG;
    }
}
#else
int rt_launch_gemm_pp(const GemmArgs&, int, hipStream_t) { return RT_ERR_BADARG; }
#endif
