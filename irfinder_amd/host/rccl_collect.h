/* rccl_collect.h — one process per GPU: the shards' results are collected into the root rank's context over RCCL.
 *
 * The recipe of include/irf.h ("Multi-GPU collection for C / C++ hosts") and INTEGRATION.md §5 as code a host links
 * against <rccl/rccl.h>.  No upstream counterpart (upstream is one process); SURVEY.md §8 e.  Every rank calls
 * irf_rccl_collect after irf_finalize; afterwards the ROOT's context answers irf_get_results / irf_write_outputs for
 * the whole sample.  Plain C (also valid C++), header-only.
 *
 *   1. JunctionCount::Directional() is a vote over ALL known junctions: all-reduce of the two sums, the decision is
 *      imposed on every rank (irf_finalize_directional);
 *   2. the fixed-shape columns: a shard's IR rows leave irf_export_device zero where it does not own the intron, so the
 *      sum of the shards IS the table (x + 0 = x, exact in fp64); the root contributes zeros (its own rows are in its
 *      context already) and adds the reduced sum with irf_import_device;
 *   3. novel junctions travel as coordinates (their packed keys depend on the shard's plane layout): five uint32
 *      columns per peer, ncclSend / ncclRecv, irf_import_junctions on the root.
 * All collectives run on `stream`; the function returns after the stream has drained.                                  */
#ifndef IRF_RCCL_COLLECT_H
#define IRF_RCCL_COLLECT_H

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/irf.h"

#define IRF_RCCL_E_HIP (-101)
#define IRF_RCCL_E_NCCL (-102)

#define IRF_RCCL_HIP(x) do { if ((x) != hipSuccess) { rc = IRF_RCCL_E_HIP; goto done; } } while (0)
#define IRF_RCCL_NCCL(x) do { if ((x) != ncclSuccess) { rc = IRF_RCCL_E_NCCL; goto done; } } while (0)
#define IRF_RCCL_IRF(x) do { rc = (x); if (rc != 0) goto done; } while (0)

static inline int irf_rccl_collect(irf_ctx* ctx, ncclComm_t comm, int rank, int nranks, int root, uint32_t dir_ratio, hipStream_t stream) {
    static const uint32_t cols[8] = {IRF_COL_IR_NONDIR_F64, IRF_COL_IR_NONDIR_U32, IRF_COL_IR_DIR_F64, IRF_COL_IR_DIR_U32,
                                     IRF_COL_SP_CNT, IRF_COL_ROI_CNT, IRF_COL_CHR_CNT, IRF_COL_KNOWN_CNT};
    int rc = 0;
    void* d_buf = NULL;
    size_t d_cap = 0;
    uint64_t* d_votes = NULL;
    uint32_t* h_j = NULL;
    uint32_t* d_j = NULL;
    uint64_t votes[2] = {0, 0};
    int c, peer;

    /* ---- 1. the strandedness vote ---- */
    IRF_RCCL_IRF(irf_get_votes(ctx, votes));
    IRF_RCCL_HIP(hipMalloc((void**)&d_votes, 2 * sizeof(uint64_t)));
    IRF_RCCL_HIP(hipMemcpyAsync(d_votes, votes, sizeof votes, hipMemcpyHostToDevice, stream));
    IRF_RCCL_NCCL(ncclAllReduce(d_votes, d_votes, 2, ncclUint64, ncclSum, comm, stream));
    IRF_RCCL_HIP(hipMemcpyAsync(votes, d_votes, sizeof votes, hipMemcpyDeviceToHost, stream));
    IRF_RCCL_HIP(hipStreamSynchronize(stream));
    IRF_RCCL_IRF(irf_finalize_directional(ctx, votes[0] > votes[1] * (uint64_t)dir_ratio ? 1 : (votes[1] > votes[0] * (uint64_t)dir_ratio ? -1 : 0)));

    /* ---- 2. fixed-shape columns: reduce(SUM) to the root ---- */
    for (c = 0; c < 8; c++) {
        const uint32_t what = cols[c];
        const int f64 = what == IRF_COL_IR_NONDIR_F64 || what == IRF_COL_IR_DIR_F64;
        uint64_t n = 0;
        size_t bytes;
        IRF_RCCL_IRF(irf_export_device(ctx, what, NULL, &n));            /* size (the same on every rank: same tables, same vote) */
        if (!n) continue;
        bytes = (size_t)n * (f64 ? 8 : 4);
        if (bytes > d_cap) {
            if (d_buf) IRF_RCCL_HIP(hipFree(d_buf));
            d_buf = NULL;
            IRF_RCCL_HIP(hipMalloc(&d_buf, bytes));
            d_cap = bytes;
        }
        if (rank == root) {
            IRF_RCCL_HIP(hipMemsetAsync(d_buf, 0, bytes, stream));        /* the root's own rows are already in its context */
        } else {
            IRF_RCCL_IRF(irf_export_device(ctx, what, d_buf, &n));         /* (returns after the copy and the row mask) */
        }
        IRF_RCCL_NCCL(ncclReduce(d_buf, d_buf, n, f64 ? ncclDouble : ncclUint32, ncclSum, root, comm, stream));
        IRF_RCCL_HIP(hipStreamSynchronize(stream));
        if (rank == root && nranks > 1) IRF_RCCL_IRF(irf_import_device(ctx, what, d_buf, n));     /* adds the peers' sum */
    }

    /* ---- 3. novel junctions as coordinates: every peer's rows to the root ---- */
    {
        uint64_t n_mine = 0;
        uint64_t* d_n = d_votes;                                          /* (reused: one count per message) */
        IRF_RCCL_IRF(irf_export_junctions(ctx, &n_mine, NULL, NULL, NULL, NULL, NULL));
        if (rank != root) {
            uint64_t n = n_mine;
            IRF_RCCL_HIP(hipMemcpyAsync(d_n, &n_mine, sizeof n_mine, hipMemcpyHostToDevice, stream));
            IRF_RCCL_NCCL(ncclSend(d_n, 1, ncclUint64, root, comm, stream));
            if (n_mine) {
                h_j = (uint32_t*)malloc((size_t)n_mine * 5 * sizeof(uint32_t));
                if (!h_j) { rc = IRF_E_NOMEM; goto done; }
                IRF_RCCL_IRF(irf_export_junctions(ctx, &n, h_j, h_j + n_mine, h_j + 2 * n_mine, h_j + 3 * n_mine, h_j + 4 * n_mine));
                IRF_RCCL_HIP(hipMalloc((void**)&d_j, (size_t)n_mine * 5 * sizeof(uint32_t)));
                IRF_RCCL_HIP(hipMemcpyAsync(d_j, h_j, (size_t)n_mine * 5 * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                IRF_RCCL_NCCL(ncclSend(d_j, (size_t)n_mine * 5, ncclUint32, root, comm, stream));
            }
            IRF_RCCL_HIP(hipStreamSynchronize(stream));
        } else {
            for (peer = 0; peer < nranks; peer++) {
                uint64_t n_peer = 0;
                if (peer == root) continue;
                IRF_RCCL_NCCL(ncclRecv(d_n, 1, ncclUint64, peer, comm, stream));
                IRF_RCCL_HIP(hipMemcpyAsync(&n_peer, d_n, sizeof n_peer, hipMemcpyDeviceToHost, stream));
                IRF_RCCL_HIP(hipStreamSynchronize(stream));
                if (!n_peer) continue;
                free(h_j); h_j = NULL;
                if (d_j) { IRF_RCCL_HIP(hipFree(d_j)); d_j = NULL; }
                h_j = (uint32_t*)malloc((size_t)n_peer * 5 * sizeof(uint32_t));
                if (!h_j) { rc = IRF_E_NOMEM; goto done; }
                IRF_RCCL_HIP(hipMalloc((void**)&d_j, (size_t)n_peer * 5 * sizeof(uint32_t)));
                IRF_RCCL_NCCL(ncclRecv(d_j, (size_t)n_peer * 5, ncclUint32, peer, comm, stream));
                IRF_RCCL_HIP(hipMemcpyAsync(h_j, d_j, (size_t)n_peer * 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                IRF_RCCL_HIP(hipStreamSynchronize(stream));
                IRF_RCCL_IRF(irf_import_junctions(ctx, n_peer, h_j, h_j + n_peer, h_j + 2 * n_peer, h_j + 3 * n_peer, h_j + 4 * n_peer));
            }
        }
    }
done:
    free(h_j);
    if (d_j) (void)hipFree(d_j);
    if (d_buf) (void)hipFree(d_buf);
    if (d_votes) (void)hipFree(d_votes);
    return rc;
}

#endif
