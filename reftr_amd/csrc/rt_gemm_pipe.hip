// rt_conv_gemm, software-pipelined LDS-DMA variants (PIPE = 1 of gemm_dma_body, rt_gemm_dma.h): fragments of K tile kt+1 are
// read into a second register set under the MFMAs of tile kt, one barrier per K tile.  Own translation unit so that it
// compiles beside rt_gemm.hip.
#include "rt_gemm_dma.h"

int rt_launch_gemm_pipe(const GemmArgs& a, int hint, hipStream_t s) {
    switch (hint) {
        // (tile, stages, min workgroups / CU, waves, PIPE)
        case 233: return launch_gemm_dma<64, 64, 3, 3, 4, 1>(a, s);
        case 252: return launch_gemm_dma<128, 128, 3, 1, 8, 1>(a, s);
        case 262: return launch_gemm_dma_dense<256, 128, 3, 1, 8, 1>(a, s);
        // small tiles for the few-row Linears (BERT at M = B * L = 320: 60 tiles of 64 x 64 leave 196 CUs idle and every busy CU's
        // load path at ~50 GB/s; 32-row / 32-column tiles put the same bytes through 2-4x the CUs): dense rows only
        case 281: return launch_gemm_dma_dense<32, 32, 3, 4, 4, 1>(a, s);
        // deep-stage form (round 4): at <= 1 workgroup per CU the K loop of a few-tile product is bound by the bytes one workgroup
        // keeps in flight ((NS - 1) K tiles), not by its MFMAs
        case 285: return launch_gemm_dma_dense<32, 32, 6, 2, 4, 1>(a, s);
#ifdef RT_LAB       // measured, not chosen by the product heuristics (LAB_NOTES.md)
        case 231: return launch_gemm_dma<64, 64, 2, 4, 4, 1>(a, s);
        case 221: return launch_gemm_dma<128, 64, 2, 2, 4, 1>(a, s);
        case 211: return launch_gemm_dma<128, 128, 2, 2, 4, 1>(a, s);
        case 251: return launch_gemm_dma<128, 128, 2, 2, 8, 1>(a, s);
        case 261: return launch_gemm_dma_dense<256, 128, 2, 1, 8, 1>(a, s);
        case 81:  return launch_gemm_dma_dense<32, 32, 3, 4, 4, 0>(a, s);
        case 282: return launch_gemm_dma_dense<32, 64, 3, 4, 4, 1>(a, s);
        case 283: return launch_gemm_dma_dense<64, 32, 3, 4, 4, 1>(a, s);
        case 284: return launch_gemm_dma_dense<32, 64, 2, 4, 4, 1>(a, s);
        case 286: return launch_gemm_dma_dense<32, 32, 8, 2, 4, 1>(a, s);
        case 287: return launch_gemm_dma_dense<32, 64, 6, 2, 4, 1>(a, s);
        case 288: return launch_gemm_dma_dense<64, 32, 6, 2, 4, 1>(a, s);
        case 234: return launch_gemm_dma_dense<64, 64, 4, 2, 4, 1>(a, s);
        case 236: return launch_gemm_dma_dense<64, 64, 6, 1, 4, 1>(a, s);
#endif
        default: return RT_ERR_BADARG;
    }
}
