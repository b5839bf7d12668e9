// mxv.hip — masked sparse-matrix x dense-vector over an arbitrary (op_mult, op_add) semiring on gfx950.
//
// Semantics: reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:56-106 (the parity oracle):
//     r[i] = select(mask[i]) ? ((init (+) p_0) (+) p_1 ...) : init,   p_k = mult(Ax[k], v[Aj[k]]), k ascending in row i
// Entry point replaced: Algo_mxv_masked_cl<T>::execute, src/opencl/cl_mxv.hpp:65-74 (not ported: the reference
// launches one workgroup-row per CSR row; here consecutive rows are grouped into row blocks that stream a
// contiguous slice of Aj/Ax, see below).
//
// Data layout in HBM: plain CSR exactly as the reference's CpuCsr/CLCsr (Ap[n_rows+1], Aj[nnz], Ax[nnz], 4-byte
// words, reference src/cpu/cpu_formats.hpp:163-175) plus the plan's blk_row[n_blocks+1] (first row of each block).
//
// Kernel `mxv_rowblock` (associative + commutative op_add, no early exit) — HBM-bound:
//   one 256-thread workgroup per row block.  A row block = consecutive rows whose first entries fall into the same
//   2048-entry window of the nnz stream (<= 1024 rows), so a block's entries are ONE contiguous slice of Aj/Ax:
//     1. Ap slice + mask slice -> LDS (row starts, per-row selected flag);
//     2. the slice of Aj/Ax is read once with 16-byte loads (4 entries per lane, 1 KiB per wave instruction),
//        products mult(Ax[k], v[Aj[k]]) are written to an LDS stage (<= 4096 entries);
//     3. per-row add-monoid fold out of LDS: a lane per short row (strict left fold from init = the reference's
//        order), a wavefront per longer row (lane-contiguous chunks + in-order segmented __shfl reduction);
//     4. a last row that overflows the stage is not staged: the whole workgroup strides it straight from HBM
//        and reduces through wave shuffles + LDS.
//   Algorithmic bytes per call: 8*nnz_sel + 4*(n_rows+1) + 4*n_rows (mask) + 4*n_rows (r) + 4*n_cols (v once).
//
// Kernel `mxv_rowseq` (any op_add, optional early exit): one lane per row, strict left fold straight from HBM with
// the reference's early-exit rule.  Exact for non-associative / non-commutative op_add.
#include "block_utils.hpp"
#include "common.hpp"
#include "ops.hpp"

#include <vector>

namespace splahip {

    constexpr uint kWin      = 2048;// nnz window that defines a row block
    constexpr uint kStage    = 4096;// LDS product stage, entries
    constexpr uint kRowsCap  = 1024;// max rows per block
    constexpr uint kTpb      = 256; // threads per workgroup
    constexpr uint kShortMax = 32;  // rows up to this length are folded by one lane
    constexpr uint kMedCap   = kStage / (kShortMax + 1) + 2;

    // ------------------------------------------------------------------------------------------------------------
    // plan construction (device side)
    // ------------------------------------------------------------------------------------------------------------

    __device__ __forceinline__ bool row_is_head(const uint* __restrict__ Ap, uint r) {
        return r == 0 || (r % kRowsCap) == 0 || (Ap[r] / kWin) != (Ap[r - 1] / kWin);
    }

    // pass 1: heads per chunk of 1024 rows (+ rows longer than the stage, for information)
    __global__ __launch_bounds__(kTpb) void plan_count_kernel(const uint* __restrict__ Ap, uint n_rows,
                                                              uint* __restrict__ chunk_heads, uint* __restrict__ n_long) {
        __shared__ uint s_tmp[kTpb / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            heads = 0, longs = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            const uint r = first + u;
            if (r < n_rows) {
                heads += row_is_head(Ap, r) ? 1u : 0u;
                longs += (Ap[r + 1] - Ap[r]) > kStage ? 1u : 0u;
            }
        }
        uint total;
        (void) block_exclusive_sum<kTpb>(heads, s_tmp, total);
        if (threadIdx.x == 0) chunk_heads[blockIdx.x] = total;
        longs = wave_sum(longs);
        if (lane_id() == 0 && longs) atomicAdd(n_long, longs);
    }

    // pass 2: exclusive scan of the per-chunk head counts (single workgroup), total -> chunk_heads[n_chunks]
    __global__ __launch_bounds__(1024) void plan_scan_kernel(uint* __restrict__ chunk_heads, uint n_chunks) {
        __shared__ uint s_tmp[1024 / kWave + 1];
        __shared__ uint s_carry;
        if (threadIdx.x == 0) s_carry = 0;
        __syncthreads();
        for (uint base = 0; base < n_chunks; base += 1024) {
            const uint i = base + threadIdx.x;
            const uint x = i < n_chunks ? chunk_heads[i] : 0;
            uint       total;
            const uint ex    = block_exclusive_sum<1024>(x, s_tmp, total);
            const uint carry = s_carry;
            if (i < n_chunks) chunk_heads[i] = carry + ex;
            __syncthreads();
            if (threadIdx.x == 0) s_carry = carry + total;
            __syncthreads();
        }
        if (threadIdx.x == 0) chunk_heads[n_chunks] = s_carry;
    }

    // pass 3: write the head rows in ascending order
    __global__ __launch_bounds__(kTpb) void plan_write_kernel(const uint* __restrict__ Ap, uint n_rows,
                                                              const uint* __restrict__ chunk_offset, uint n_chunks,
                                                              uint* __restrict__ blk_row) {
        __shared__ uint s_tmp[kTpb / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        bool            head[4];
        uint            heads = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            const uint r = first + u;
            head[u]      = r < n_rows && row_is_head(Ap, r);
            heads += head[u] ? 1u : 0u;
        }
        uint total;
        uint pos = chunk_offset[blockIdx.x] + block_exclusive_sum<kTpb>(heads, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (head[u]) blk_row[pos++] = first + u;
        if (blockIdx.x == 0 && threadIdx.x == 0) blk_row[chunk_offset[n_chunks]] = n_rows;
    }

    // ------------------------------------------------------------------------------------------------------------
    // row-block kernel
    // ------------------------------------------------------------------------------------------------------------

    // which row of the block's LDS Ap slice holds entry k: last i with s_ap[i] <= k  (k inside the block's slice)
    __device__ __forceinline__ uint find_row(const uint* s_ap, uint nr, uint k) {
        uint lo = 0, hi = nr;// invariant: s_ap[lo] <= k < s_ap[hi]
        while (hi - lo > 1) {
            const uint mid = (lo + hi) >> 1;
            if (s_ap[mid] <= k) lo = mid;
            else
                hi = mid;
        }
        return lo;
    }

    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(kTpb) void mxv_rowblock_kernel(const uint* __restrict__ blk_row,
                                                                const uint* __restrict__ Ap,
                                                                const uint* __restrict__ Aj,
                                                                const uint* __restrict__ Ax,
                                                                const T* __restrict__ mask,
                                                                const T* __restrict__ v,
                                                                T* __restrict__ r,
                                                                Mult mult, Add add, int op_select, T init,
                                                                uint nnz, int vec_ok) {
        __shared__ uint          s_ap[kRowsCap + 1];
        __shared__ T             s_prod[kStage + 4];
        __shared__ unsigned char s_sel[kRowsCap];
        __shared__ uint          s_med[kMedCap];
        __shared__ uint          s_nmed;
        __shared__ T             s_wpart[kTpb / kWave];
        __shared__ int           s_wvalid[kTpb / kWave];

        const uint tid = threadIdx.x;
        const uint r0  = blk_row[blockIdx.x];
        const uint nr  = blk_row[blockIdx.x + 1] - r0;

        // ---- 1. row starts and mask -> LDS
        if (tid == 0) s_nmed = 0;
        for (uint i = tid; i <= nr; i += kTpb) s_ap[i] = Ap[r0 + i];
        uint my_rows = 0, my_sel = 0;
        for (uint i = tid; i < nr; i += kTpb) {
            const bool s = eval_sel(op_select, mask[r0 + i]);
            s_sel[i]     = s ? 1 : 0;
            my_rows += 1;
            my_sel += s ? 1u : 0u;
        }
        const int any_sel = __syncthreads_or((int) my_sel);
        if (!any_sel) {// nothing selected: every row gets init, no matrix entry is touched
            for (uint i = tid; i < nr; i += kTpb) r[r0 + i] = init;
            return;
        }
        const int all_sel = __syncthreads_and(my_sel == my_rows);

        const uint base    = s_ap[0];
        const uint end     = s_ap[nr];
        const uint a0      = base & ~3u;              // stage origin, 16-byte aligned in Aj/Ax
        const bool peel    = (end - a0) > kStage;     // the last row overflows the stage: handled in step 4
        const uint nr_body = peel ? nr - 1 : nr;
        const uint bend    = s_ap[nr_body];           // end of the staged slice
        const uint nq      = bend > a0 ? (bend - a0 + 3) >> 2 : 0;// quads to stage, <= 1024

        // ---- 2. stream the slice once, stage products in LDS
        {
            uint4 jq[4], xq[4];
            bool  act[4];
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                const uint qi = tid + u * kTpb;
                act[u]        = qi < nq;
                if (act[u] && !all_sel) {
                    // skip quads that only cover unselected rows
                    const uint k0  = a0 + 4 * qi;
                    bool       need = false;
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        const uint k = k0 + e;
                        if (k >= base && k < bend) need |= s_sel[find_row(s_ap, nr, k)] != 0;
                    }
                    act[u] = need;
                }
                if (act[u]) {
                    const uint k0 = a0 + 4 * qi;
                    if (vec_ok && k0 + 4 <= nnz) {
                        jq[u] = *reinterpret_cast<const uint4*>(Aj + k0);
                        xq[u] = *reinterpret_cast<const uint4*>(Ax + k0);
                    } else {
                        uint jj[4] = {0, 0, 0, 0}, xx[4] = {0, 0, 0, 0};
#pragma unroll
                        for (uint e = 0; e < 4; ++e) {
                            const uint k = k0 + e;
                            if (k >= base && k < bend) {
                                jj[e] = Aj[k];
                                xx[e] = Ax[k];
                            }
                        }
                        jq[u] = make_uint4(jj[0], jj[1], jj[2], jj[3]);
                        xq[u] = make_uint4(xx[0], xx[1], xx[2], xx[3]);
                    }
                }
            }
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                if (act[u]) {
                    const uint qi = tid + u * kTpb;
                    const uint k0 = a0 + 4 * qi;
                    const uint jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
                    const uint xx[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
                    T          vv[4];
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        const uint k = k0 + e;
                        vv[e]        = (k >= base && k < bend) ? v[jj[e]] : T(0);
                    }
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        const uint k = k0 + e;
                        if (k >= base && k < bend) s_prod[4 * qi + e] = mult(from_bits<T>(xx[e]), vv[e]);
                    }
                }
            }
        }
        __syncthreads();

        // ---- 3a. a lane per short row: strict left fold from init, the reference's order
        for (uint i = tid; i < nr_body; i += kTpb) {
            T out = init;
            if (s_sel[i]) {
                const uint s = s_ap[i] - a0, e = s_ap[i + 1] - a0;
                if (e - s <= kShortMax) {
                    for (uint k = s; k < e; ++k) out = add(out, s_prod[k]);
                } else {
                    s_med[atomicAdd(&s_nmed, 1u)] = i;
                    continue;
                }
            }
            r[r0 + i] = out;
        }
        __syncthreads();

        // ---- 3b. a wavefront per longer staged row: lane-contiguous chunks (odd length: conflict-free LDS reads),
        //          then an in-order segmented shuffle reduction
        {
            const uint lane = lane_id();
            const uint nmed = s_nmed;
            for (uint t = wave_id(); t < nmed; t += kTpb / kWave) {
                const uint i   = s_med[t];
                const uint s   = s_ap[i] - a0, e = s_ap[i + 1] - a0;
                const uint c   = ((e - s + kWave - 1) / kWave) | 1u;
                const uint ks  = s + lane * c;
                const uint ke  = min(ks + c, e);
                bool       valid = ks < e;
                T          part  = init;
                if (valid) {
                    part = s_prod[ks];
                    for (uint k = ks + 1; k < ke; ++k) part = add(part, s_prod[k]);
                }
                wave_reduce_ordered(part, valid, add);
                if (lane == 0) r[r0 + i] = valid ? add(init, part) : init;
            }
        }

        // ---- 4. the peeled long row: all 256 lanes stride it from HBM (16-byte loads), reduce via shuffles + LDS
        if (peel) {
            const uint i = nr - 1;
            if (!s_sel[i]) {
                if (tid == 0) r[r0 + i] = init;
                return;
            }
            const uint s = s_ap[i], e = s_ap[i + 1];
            const uint q_lo = s >> 2, q_hi = (e + 3) >> 2;
            T          part  = init;
            bool       valid = false;
#pragma unroll 2
            for (uint q = q_lo + tid; q < q_hi; q += kTpb) {
                const uint k0 = q << 2;
                uint       jj[4], xx[4];
                if (vec_ok && k0 >= s && k0 + 4 <= e) {
                    const uint4 jv = *reinterpret_cast<const uint4*>(Aj + k0);
                    const uint4 xv = *reinterpret_cast<const uint4*>(Ax + k0);
                    jj[0] = jv.x, jj[1] = jv.y, jj[2] = jv.z, jj[3] = jv.w;
                    xx[0] = xv.x, xx[1] = xv.y, xx[2] = xv.z, xx[3] = xv.w;
                    T vv[4];
#pragma unroll
                    for (uint u = 0; u < 4; ++u) vv[u] = v[jj[u]];
#pragma unroll
                    for (uint u = 0; u < 4; ++u) {
                        const T p = mult(from_bits<T>(xx[u]), vv[u]);
                        part      = valid ? add(part, p) : p;
                        valid     = true;
                    }
                } else {
#pragma unroll
                    for (uint u = 0; u < 4; ++u) {
                        const uint k = k0 + u;
                        if (k >= s && k < e) {
                            const T p = mult(from_bits<T>(Ax[k]), v[Aj[k]]);
                            part      = valid ? add(part, p) : p;
                            valid     = true;
                        }
                    }
                }
            }
            wave_reduce_ordered(part, valid, add);
            if (lane_id() == 0) {
                s_wpart[wave_id()]  = part;
                s_wvalid[wave_id()] = valid ? 1 : 0;
            }
            __syncthreads();
            if (tid == 0) {
                T sum = init;
                for (uint w = 0; w < kTpb / kWave; ++w)
                    if (s_wvalid[w]) sum = add(sum, s_wpart[w]);
                r[r0 + i] = sum;
            }
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // sequential-per-row kernel: exact reference order for ANY op_add, with the reference's early exit
    // ------------------------------------------------------------------------------------------------------------
    template<typename T, class Mult, class Add, bool EARLY>
    __global__ __launch_bounds__(kTpb) void mxv_rowseq_kernel(uint n_rows,
                                                              const uint* __restrict__ Ap,
                                                              const uint* __restrict__ Aj,
                                                              const uint* __restrict__ Ax,
                                                              const T* __restrict__ mask,
                                                              const T* __restrict__ v,
                                                              T* __restrict__ r,
                                                              Mult mult, Add add, int op_select, T init) {
        const uint i = blockIdx.x * kTpb + threadIdx.x;
        if (i >= n_rows) return;
        T sum = init;
        if (eval_sel(op_select, mask[i])) {
            const uint e = Ap[i + 1];
            for (uint k = Ap[i]; k < e; ++k) {
                sum = add(sum, mult(from_bits<T>(Ax[k]), v[Aj[k]]));
                if (EARLY && sum != init) break;
            }
        }
        r[i] = sum;
    }

}// namespace splahip

using namespace splahip;

struct splahip_plan_s {
    uint32_t  n_rows   = 0;
    uint32_t  nnz      = 0;
    uint32_t  n_blocks = 0;
    uint32_t  n_long   = 0;
    uint32_t* d_blk_row = nullptr;// n_blocks + 1
};

namespace {

    enum class Variant { RowBlockStatic,
                         RowBlockDynamic,
                         RowSeq,
                         RowSeqEarly };

    // semirings with an ahead-of-time instantiated row-block kernel
    bool static_semiring(int type, int op_mult, int op_add) {
        if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS) return true;
        if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN) return true;
        if (op_mult == SPLAHIP_LAND && op_add == SPLAHIP_LOR) return true;
        if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR) return type != SPLAHIP_FLOAT;
        return false;
    }

    Variant pick_variant(int type, int op_mult, int op_add, int early_exit) {
        if (early_exit) return Variant::RowSeqEarly;
        if (!binop_is_associative(op_add) || !binop_is_commutative(op_add)) return Variant::RowSeq;
        return static_semiring(type, op_mult, op_add) ? Variant::RowBlockStatic : Variant::RowBlockDynamic;
    }

    template<typename T, class Mult, class Add>
    int launch_rowblock(splahip_plan plan, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const void* mask,
                        const void* v, void* r, int op_mult, int op_add, int op_select, uint32_t init_bits, hipStream_t st) {
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(Aj) | reinterpret_cast<uintptr_t>(Ax)) & 15u) == 0;
        hipLaunchKernelGGL((mxv_rowblock_kernel<T, Mult, Add>), dim3(plan->n_blocks), dim3(kTpb), 0, st,
                           plan->d_blk_row, Ap, Aj, static_cast<const uint32_t*>(Ax), static_cast<const T*>(mask),
                           static_cast<const T*>(v), static_cast<T*>(r), Mult(op_mult), Add(op_add), op_select,
                           from_bits<T>(init_bits), plan->nnz, vec_ok ? 1 : 0);
        SPLAHIP_TRY(hipGetLastError());
        return SPLAHIP_OK;
    }

    template<typename T, bool EARLY>
    int launch_rowseq(uint32_t n_rows, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const void* mask,
                      const void* v, void* r, int op_mult, int op_add, int op_select, uint32_t init_bits, hipStream_t st) {
        hipLaunchKernelGGL((mxv_rowseq_kernel<T, DynBin, DynBin, EARLY>), dim3((n_rows + kTpb - 1) / kTpb), dim3(kTpb), 0, st,
                           n_rows, Ap, Aj, static_cast<const uint32_t*>(Ax), static_cast<const T*>(mask),
                           static_cast<const T*>(v), static_cast<T*>(r), DynBin(op_mult), DynBin(op_add), op_select,
                           from_bits<T>(init_bits));
        SPLAHIP_TRY(hipGetLastError());
        return SPLAHIP_OK;
    }

    template<typename T>
    int dispatch_typed(splahip_plan plan, uint32_t n_rows, const uint32_t* Ap, const uint32_t* Aj, const void* Ax,
                       const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                       uint32_t init_bits, int early_exit, hipStream_t st) {
        constexpr int type = TypeId<T>::value;
        switch (pick_variant(type, op_mult, op_add, early_exit)) {
            case Variant::RowSeqEarly:
                return launch_rowseq<T, true>(n_rows, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case Variant::RowSeq:
                return launch_rowseq<T, false>(n_rows, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case Variant::RowBlockDynamic:
                return launch_rowblock<T, DynBin, DynBin>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case Variant::RowBlockStatic:
                if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS)
                    return launch_rowblock<T, StaBin<SPLAHIP_MULT>, StaBin<SPLAHIP_PLUS>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
                if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN)
                    return launch_rowblock<T, StaBin<SPLAHIP_PLUS>, StaBin<SPLAHIP_MIN>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
                if (op_mult == SPLAHIP_LAND && op_add == SPLAHIP_LOR)
                    return launch_rowblock<T, StaBin<SPLAHIP_LAND>, StaBin<SPLAHIP_LOR>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
                if constexpr (type != SPLAHIP_FLOAT) {
                    if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR)
                        return launch_rowblock<T, StaBin<SPLAHIP_BAND>, StaBin<SPLAHIP_BOR>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
                }
                break;
        }
        return fail(SPLAHIP_NOT_IMPLEMENTED, "no kernel variant for this semiring");
    }

}// namespace

extern "C" {

int splahip_csr_plan_create(splahip_plan* out, uint32_t n_rows, uint32_t nnz, const uint32_t* d_Ap, void* stream) {
    SPLAHIP_REQUIRE(out != nullptr, "plan is null");
    SPLAHIP_REQUIRE(d_Ap != nullptr || n_rows == 0, "d_Ap is null");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);

    auto* plan   = new splahip_plan_s();
    plan->n_rows = n_rows;
    plan->nnz    = nnz;
    if (n_rows == 0) {
        *out = plan;
        return SPLAHIP_OK;
    }

    const uint32_t n_chunks = (n_rows + 1023u) / 1024u;
    uint32_t*      d_chunk  = nullptr;// n_chunks + 1 offsets, then 1 word n_long
    auto           cleanup  = [&]() {
        if (d_chunk) (void) hipFree(d_chunk);
    };
    if (hipMalloc(reinterpret_cast<void**>(&d_chunk), sizeof(uint32_t) * (size_t) (n_chunks + 2)) != hipSuccess) {
        delete plan;
        return fail(SPLAHIP_ERROR, "hipMalloc failed for plan scratch");
    }
    uint32_t* d_nlong = d_chunk + n_chunks + 1;
    hipError_t e      = hipMemsetAsync(d_nlong, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(plan_count_kernel, dim3(n_chunks), dim3(kTpb), 0, s, d_Ap, n_rows, d_chunk, d_nlong);
        hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, s, d_chunk, n_chunks);
        e = hipGetLastError();
    }
    uint32_t h_tail[2] = {0, 0};// {n_blocks, n_long}
    if (e == hipSuccess) e = hipMemcpyAsync(h_tail, d_chunk + n_chunks, sizeof(h_tail), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
        plan->n_blocks = h_tail[0];
        plan->n_long   = h_tail[1];
        e              = hipMalloc(reinterpret_cast<void**>(&plan->d_blk_row), sizeof(uint32_t) * (size_t) (plan->n_blocks + 1));
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(plan_write_kernel, dim3(n_chunks), dim3(kTpb), 0, s, d_Ap, n_rows, d_chunk, n_chunks, plan->d_blk_row);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) {
        if (plan->d_blk_row) (void) hipFree(plan->d_blk_row);
        delete plan;
        return fail(SPLAHIP_ERROR, "plan construction failed: %s", hipGetErrorString(e));
    }
    *out = plan;
    return SPLAHIP_OK;
}

int splahip_csr_plan_destroy(splahip_plan plan) {
    if (!plan) return SPLAHIP_OK;
    if (plan->d_blk_row) (void) hipFree(plan->d_blk_row);
    delete plan;
    return SPLAHIP_OK;
}

int splahip_csr_plan_info(splahip_plan plan, uint32_t* n_blocks, uint32_t* n_long) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    if (n_blocks) *n_blocks = plan->n_blocks;
    if (n_long) *n_long = plan->n_long;
    return SPLAHIP_OK;
}

const char* splahip_mxv_variant(int type, int op_mult, int op_add, int early_exit) {
    switch (pick_variant(type, op_mult, op_add, early_exit)) {
        case Variant::RowBlockStatic: return "mxv_rowblock<static semiring>";
        case Variant::RowBlockDynamic: return "mxv_rowblock<runtime ops>";
        case Variant::RowSeq: return "mxv_rowseq";
        case Variant::RowSeqEarly: return "mxv_rowseq<early_exit>";
    }
    return "?";
}

int splahip_mxv_masked(splahip_plan plan, int type, uint32_t n_rows, uint32_t n_cols,
                       const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                       const void* d_mask, const void* d_v, void* d_r,
                       int op_mult, int op_add, int op_select, uint32_t init_bits, int early_exit, void* stream) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    SPLAHIP_REQUIRE(plan->n_rows == n_rows, "plan was built for a different matrix (n_rows)");
    SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
    SPLAHIP_REQUIRE(binop_valid_for(op_mult, type), "op_multiply not defined for this type");
    SPLAHIP_REQUIRE(binop_valid_for(op_add, type), "op_add not defined for this type");
    SPLAHIP_REQUIRE(op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT, "unknown op_select");
    (void) n_cols;
    if (n_rows == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_Ap && d_mask && d_r, "null device pointer");
    SPLAHIP_REQUIRE(plan->nnz == 0 || (d_Aj && d_Ax && d_v), "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    switch (type) {
        case SPLAHIP_INT:
            return dispatch_typed<int32_t>(plan, n_rows, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, early_exit, s);
        case SPLAHIP_UINT:
            return dispatch_typed<uint32_t>(plan, n_rows, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, early_exit, s);
        default:
            return dispatch_typed<float>(plan, n_rows, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, early_exit, s);
    }
}
}
