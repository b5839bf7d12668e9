// sell.hip — construction of the sliced-ELL + hot-column copy of a device CSR matrix and the launch of the masked SpMV over
// it (kernels and the rationale: mxv_sell.hpp).  Everything is built on the device: a matrix that lives in HBM never
// visits the host.  The copy belongs to the plan (splahip_csr_plan_attach_sell) and must be rebuilt when the matrix changes.
#include "block_utils.hpp"
#include "common.hpp"
#include "mxv_sell.hpp"
#include "ops.hpp"
#include "plan.hpp"

#include <chrono>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

using namespace splahip;
using SC = SellCfg;

namespace {

    // ---- plan-time kernels ----------------------------------------------------------------------------------------------
    // per row: chunks it contributes, chunks that need a partial slot (rows longer than one chunk), "is long" flag
    constexpr uint kXBlock = kWave * SC::kC;// entries of a row-exclusive slice

    __global__ __launch_bounds__(256) void sell_rowinfo_kernel(const uint* __restrict__ Ap, uint n_rows, uint* __restrict__ vcount, uint* __restrict__ pcount,
                                                               uint* __restrict__ lflag, uint* __restrict__ xcount) {
        const uint row = blockIdx.x * 256 + threadIdx.x;
        if (row > n_rows) return;
        uint c = 0, l = 0, pc = 0, xs = 0;
        if (row < n_rows) {
            const uint len = Ap[row + 1] - Ap[row];
            l              = len > SC::kC ? 1u : 0u;
            if (len > SC::kXL) {// row-exclusive slices; a partial slot per non-empty lane
                xs = (len + kXBlock - 1) / kXBlock;
                pc = (xs - 1) * kWave + min(kWave, len - (xs - 1) * kXBlock);
                l  = 1;
            } else {
                c  = (len + SC::kC - 1) / SC::kC;
                pc = l ? c : 0u;
            }
        }
        vcount[row] = c;// entry n_rows = 0: the scans run over n_rows + 1 values so their last output is the total
        pcount[row] = pc;
        lflag[row]  = l;
        xcount[row] = xs;
    }
    // per row: its chunks in row order (the stable sort below keeps a long row's chunks adjacent and ascending)
    __global__ __launch_bounds__(256) void sell_emit_kernel(const uint* __restrict__ Ap, uint n_rows, const uint* __restrict__ vfirst, const uint* __restrict__ pfirst,
                                                            const uint* __restrict__ lfirst, uint* __restrict__ q_key, uint* __restrict__ q_id, uint* __restrict__ q_row,
                                                            uint* __restrict__ q_len, uint* __restrict__ q_k0, uint* __restrict__ q_slot, uint* __restrict__ lr_row,
                                                            uint* __restrict__ lr_first, const uint* __restrict__ xfirst, uint* __restrict__ vrow,
                                                            uint* __restrict__ vlen, uint* __restrict__ vk0, uint* __restrict__ vslot) {
        const uint row = blockIdx.x * 256 + threadIdx.x;
        if (row >= n_rows) return;
        const uint k0  = Ap[row];
        const uint len = Ap[row + 1] - k0;
        if (len == 0) return;
        const bool is_long = len > SC::kC;
        if (len > SC::kXL) {// row-exclusive slices go straight to the head of the virtual-row arrays (row order; taken by width via sorder)
            const uint xs = (len + kXBlock - 1) / kXBlock;
            for (uint S = 0; S < xs; ++S) {
                const uint rem = min(kXBlock, len - S * kXBlock);// entries of this slice's block
                for (uint l = 0; l < kWave; ++l) {
                    const uint p = (xfirst[row] + S) * kWave + l;
                    vrow[p]      = row | SC::kPartial;
                    vlen[p]      = l < rem ? (rem - l + kWave - 1) / kWave : 0u;
                    vk0[p]       = k0 + S * kXBlock + l;
                    vslot[p]     = pfirst[row] + S * kWave + l;// lanes >= rem hold nothing and own no slot
                }
            }
            lr_row[lfirst[row]]   = row;
            lr_first[lfirst[row]] = pfirst[row];
            return;
        }
        const uint nch     = (len + SC::kC - 1) / SC::kC;
        const uint q0      = vfirst[row];
        const uint p0      = pfirst[row];
        for (uint c = 0; c < nch; ++c) {
            const uint clen = min(SC::kC, len - c * SC::kC);
            const uint q    = q0 + c;
            q_key[q]        = SC::kC - clen;// ascending key = descending length
            q_id[q]         = q;
            q_row[q]        = row | (is_long ? SC::kPartial : 0u);
            q_len[q]        = clen;
            q_k0[q]         = k0 + c * SC::kC;
            q_slot[q]       = is_long ? p0 + c : 0u;
        }
        if (is_long) {
            lr_row[lfirst[row]]   = row;
            lr_first[lfirst[row]] = p0;
        }
    }
    // the sorted chunks follow the row-exclusive slices: virtual rows [first, n_vrows)
    __global__ __launch_bounds__(256) void sell_permute_kernel(uint n_chunks, uint first, uint n_vrows, const uint* __restrict__ order, const uint* __restrict__ q_row,
                                                               const uint* __restrict__ q_len, const uint* __restrict__ q_k0, const uint* __restrict__ q_slot,
                                                               uint* __restrict__ vrow, uint* __restrict__ vlen, uint* __restrict__ vk0, uint* __restrict__ vslot) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        const uint p = first + i;
        if (p >= n_vrows) return;
        if (i < n_chunks) {
            const uint q = order[i];
            vrow[p]      = q_row[q];
            vlen[p]      = q_len[q];
            vk0[p]       = q_k0[q];
            vslot[p]     = q_slot[q];
        } else {// padding lanes of the last slice
            vrow[p] = 0x7FFFFFFFu;
            vlen[p] = vk0[p] = vslot[p] = 0;
        }
    }
    __global__ __launch_bounds__(256) void sell_width_kernel(const uint* __restrict__ vlen, uint n_slices, uint* __restrict__ slots, uint* __restrict__ wkey,
                                                             uint* __restrict__ sid) {
        const uint s = blockIdx.x * 256 + threadIdx.x;
        if (s <= n_slices) slots[s] = s < n_slices ? vlen[s * kWave] * kWave : 0u;// lane 0 holds the slice's width
        if (s < n_slices) {
            wkey[s] = SC::kC - vlen[s * kWave];// ascending key = descending width
            sid[s]  = s;
        }
    }
    // column reference counts -> the n_hot most referenced columns get an LDS slot
    // (a workgroup counts a span of kCountSpan entries into a direct-mapped LDS table of (column, count) pairs first — a column that finds its slot taken by
    // another goes to memory at once — and adds the table's pairs to memory at the end: a hub column of a scale-free matrix costs one memory atomic per
    // workgroup instead of one per reference, which is what serialised the plain form: 7 ms for 65 M entries)
    constexpr uint kCountSpan = 16384, kCountSlots = 8192;
    // stride > 1: only every stride-th entry is counted.  The counts only RANK the columns (which 40 000 get an LDS slot, in which order the cold list goes): any
    // ranking gives a valid copy, and a 1-in-8 sample of 65 M entries ranks the hot columns like the full count does — at 8 M memory atomics instead of 65 M (the
    // full count runs at the chip's ~11 G atomics/s whatever their addresses: 5.7 ms at R-MAT 22 even with the hub columns kept in LDS)
    __global__ __launch_bounds__(256) void sell_col_count_kernel(const uint* __restrict__ Aj, uint nnz, uint stride, uint* __restrict__ count) {
        __shared__ uint tag[kCountSlots], cnt[kCountSlots];
        for (uint i = threadIdx.x; i < kCountSlots; i += 256) tag[i] = 0xFFFFFFFFu, cnt[i] = 0u;
        __syncthreads();
        const uint n_samples = (nnz + stride - 1) / stride;
        const uint k0 = blockIdx.x * kCountSpan, k1 = min(n_samples, k0 + kCountSpan);
        for (uint k = k0 + threadIdx.x; k < k1; k += 256) {
            const uint c    = Aj[(size_t) k * stride];
            const uint slot = (c * 2654435761u) >> 19;// 13 bits
            const uint seen = atomicCAS(&tag[slot], 0xFFFFFFFFu, c);
            if (seen == 0xFFFFFFFFu || seen == c) atomicAdd(&cnt[slot], 1u);
            else
                atomicAdd(&count[c], 1u);
        }
        __syncthreads();
        for (uint i = threadIdx.x; i < kCountSlots; i += 256)
            if (cnt[i] != 0u) atomicAdd(&count[tag[i]], cnt[i]);
    }
    __global__ __launch_bounds__(256) void sell_col_key_kernel(const uint* __restrict__ count, uint n_cols, uint* __restrict__ key, uint* __restrict__ col) {
        const uint j = blockIdx.x * 256 + threadIdx.x;
        if (j < n_cols) {
            key[j] = ~count[j];// ascending ~count = descending count; the stable sort keeps ascending column order among ties
            col[j] = j;
        }
    }
    // rank_of[column] = position in the descending reference-count order; the first n_hot ranks are the LDS value slots
    __global__ __launch_bounds__(256) void sell_rank_kernel(const uint* __restrict__ sorted_col, uint n_cols, uint n_hot, uint* __restrict__ rank_of,
                                                            uint* __restrict__ hot_cols) {
        const uint s = blockIdx.x * 256 + threadIdx.x;
        if (s < n_cols) {
            rank_of[sorted_col[s]] = s;
            if (s < n_hot) hot_cols[s] = sorted_col[s];
        }
    }
    // one wave per slice: lane l copies chunk l into the slice's column-major block, re-encoding hot columns; the rest of the
    // lane's column is padding that reads slot 0 of the hot table and multiplies by a zero word (never folded)
    __global__ __launch_bounds__(256) void sell_fill_kernel(uint n_slices, uint n_xslices, const uint* __restrict__ off, const uint* __restrict__ vlen, const uint* __restrict__ vk0,
                                                            const uint* __restrict__ Aj, const uint* __restrict__ Ax, const uint* __restrict__ rank_of, uint n_hot,
                                                            uint* __restrict__ Aj2, uint* __restrict__ Aj2r, uint* __restrict__ Ax2) {
        const uint s = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
        if (s >= n_slices) return;
        const uint lane  = lane_id();
        const uint base  = off[s];
        const uint width = (off[s + 1] - base) / kWave;
        const uint len    = vlen[s * kWave + lane];
        const uint k0     = vk0[s * kWave + lane];
        const uint stride = s < n_xslices ? kWave : 1u;// row-exclusive slice: the block in CSR order, lane-strided
        for (uint e = 0; e < width; ++e) {
            uint j = SC::kHotFlag, rank = SC::kBitsLds, x = 0;
            if (e < len) {
                const uint col = Aj[k0 + e * stride];
                rank           = rank_of[col];
                j              = rank < n_hot ? (rank | SC::kHotFlag) : col;
                x              = Ax[k0 + e * stride];
            }
            Aj2[base + e * kWave + lane]  = j;
            Aj2r[base + e * kWave + lane] = SC::bits_code(rank);// padding and colder columns (their entries are in the cold list): code 0
            Ax2[base + e * kWave + lane]  = x;
        }
    }
    __global__ __launch_bounds__(256) void sell_tail_kernel(uint first, uint count, uint* __restrict__ Aj2, uint* __restrict__ Aj2r, uint* __restrict__ Ax2) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k < count) {
            Aj2[first + k]  = SC::kHotFlag;
            Aj2r[first + k] = 0;
            Ax2[first + k]  = 0;
        }
    }

    // cold list: entries whose column rank has no LDS-resident bit
    __global__ __launch_bounds__(256) void sell_cold_flag_kernel(const uint* __restrict__ Aj, uint nnz, const uint* __restrict__ rank_of, uint* __restrict__ flag) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k <= nnz) flag[k] = (k < nnz && rank_of[Aj[k]] >= SC::kBitsLds) ? 1u : 0u;
    }
    __global__ __launch_bounds__(256) void sell_cold_emit_kernel(const uint* __restrict__ Ap, uint n_rows, const uint* __restrict__ Aj, const uint* __restrict__ Ax,
                                                                 uint nnz, const uint* __restrict__ rank_of, const uint* __restrict__ pos,
                                                                 uint* __restrict__ cold_rank, unsigned long long* __restrict__ cold_entry) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k >= nnz) return;
        const uint rank = rank_of[Aj[k]];
        if (rank < SC::kBitsLds) return;
        uint lo = 0, hi = n_rows;// last row with Ap[row] <= k
        while (hi - lo > 1) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (Ap[mid] <= k) lo = mid;
            else
                hi = mid;
        }
        cold_rank[pos[k]]  = rank;
        cold_entry[pos[k]] = ((unsigned long long) lo << 32) | Ax[k];
    }

    constexpr uint kTail = 4 * SC::kU * kWave;// read-ahead of the last slice's group prefetch

    struct Scratch {
        std::vector<void*> ptrs;
        ~Scratch() {
            for (void* p : ptrs)
                if (p) (void) hipFree(p);
        }
        hipError_t alloc(uint32_t** out, size_t words) {
            void*      p = nullptr;
            hipError_t e = device_malloc(&p, sizeof(uint32_t) * (words ? words : 1));
            if (e == hipSuccess) ptrs.push_back(p);
            *out = static_cast<uint32_t*>(p);
            return e;
        }
    };

    hipError_t exclusive_scan_u32(uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
        size_t     bytes = 0;
        hipError_t e     = rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
        if (e != hipSuccess) return e;
        void* tmp = nullptr;
        e         = hipMalloc(&tmp, bytes ? bytes : 4);
        if (e != hipSuccess) return e;
        e = rocprim::exclusive_scan(tmp, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void) hipFree(tmp);
        return e;
    }
    hipError_t stable_sort_pairs(uint32_t* key, uint32_t* key_s, uint32_t* val, uint32_t* val_s, size_t n, unsigned bits, hipStream_t s) {
        size_t     bytes = 0;
        hipError_t e     = rocprim::radix_sort_pairs(nullptr, bytes, key, key_s, val, val_s, n, 0, bits, s);
        if (e != hipSuccess) return e;
        void* tmp = nullptr;
        e         = hipMalloc(&tmp, bytes ? bytes : 4);
        if (e != hipSuccess) return e;
        e = rocprim::radix_sort_pairs(tmp, bytes, key, key_s, val, val_s, n, 0, bits, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void) hipFree(tmp);
        return e;
    }

    uint compute_units() {
        static uint cus = 0;
        if (!cus) {
            int             dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = (uint) prop.multiProcessorCount;
            if (!cus) cus = 256;
        }
        return cus;
    }

    // 0 annihilates op_mult and is op_add's identity: an entry whose v value is 0 cannot change the fold (mxv_sell.hpp, BITS form)
    // any negative / NaN value, and the largest value, of the copy's value array (padding words are 0)
    template<typename T>
    __global__ __launch_bounds__(256) void sell_ax_stats_kernel(uint n, const T* __restrict__ x, uint* __restrict__ out) {
        uint bad = 0, mx = 0;
        for (uint k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
            const T a = x[k];
            if constexpr (std::is_same<T, uint32_t>::value) mx = max(mx, a);
            else {
                if (a < T(0) || a != a) bad = 1;
                else if constexpr (std::is_same<T, int32_t>::value)
                    mx = max(mx, (uint) a);
            }
        }
        bad = wave_max(bad);
        mx  = wave_max(mx);
        if (lane_id() == 0) {
            if (bad) atomicOr(&out[0], 1u);
            atomicMax(&out[1], mx);
        }
    }
    // smallest and largest value bits of the CSR value array (equal <=> every stored value is the same word)
    __global__ __launch_bounds__(256) void sell_ax_range_kernel(uint n, const uint* __restrict__ x, uint* __restrict__ out) {
        uint lo = 0xFFFFFFFFu, hi = 0;
        for (uint k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
            const uint a = x[k];
            lo = min(lo, a), hi = max(hi, a);
        }
        lo = ~wave_max(~lo);
        hi = wave_max(hi);
        if (lane_id() == 0) {
            atomicMin(&out[0], lo);
            atomicMax(&out[1], hi);
        }
    }
    template<typename T>
    int ax_stats(const splahip_sell_s* g, hipStream_t st) {
        if (g->ax_type == TypeId<T>::value) return SPLAHIP_OK;
        DevMem out;
        SPLAHIP_TRY(out.alloc(st, 8));
        SPLAHIP_TRY(hipMemsetAsync(out.p, 0, 8, st));
        hipLaunchKernelGGL((sell_ax_stats_kernel<T>), dim3(1024), dim3(256), 0, st, g->n_slots, reinterpret_cast<const T*>(g->d_Ax2), out.as<uint>());
        uint h[2] = {1, 0};
        SPLAHIP_TRY(hipMemcpyAsync(h, out.p, 8, hipMemcpyDeviceToHost, st));
        SPLAHIP_TRY(hipStreamSynchronize(st));
        g->ax_bad = h[0], g->ax_max = h[1], g->ax_type = TypeId<T>::value;
        return SPLAHIP_OK;
    }
    // min-plus family: is "v[j] == init contributes nothing" true for this matrix and this init?  a + init >= init for every matrix value a
    // (none negative, no NaN) and, for the integer types, without wrap-around
    template<typename T>
    bool min_plus_skippable(const splahip_sell_s* g, uint32_t init_bits) {
        if (g->ax_bad) return false;
        if constexpr (std::is_same<T, float>::value) {
            const float init = from_bits<float>(init_bits);
            return init == init;
        } else if constexpr (std::is_same<T, int32_t>::value)
            return (long long) from_bits<int32_t>(init_bits) + (long long) g->ax_max <= 0x7FFFFFFFll;
        else
            return (unsigned long long) init_bits + (unsigned long long) g->ax_max <= 0xFFFFFFFFull;
    }
    bool uniform_disabled() {// SPLAHIP_SELL_UNIFORM=0: stream the value words even when they are all equal (diagnostic / A-B)
        const char* e = knob_env("SPLAHIP_SELL_UNIFORM");
        return e && e[0] == '0';
    }
    int bits_force() {// SPLAHIP_SELL_BITS: 0 = never, 1 / unset = the device statistics decide, 2 = always (where the semiring allows it)
        const char* e = knob_env("SPLAHIP_SELL_BITS");
        return e ? std::atoi(e) : 1;
    }

    template<typename T, class Mult, class Add>
    int launch_sell(const splahip_plan_s* plan, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                    uint32_t init_bits, hipStream_t st) {
        const splahip_sell_s* g    = plan->sell;
        const uint            grid = compute_units();
        const int             mode = bits_force();
        const int             family = mode != 0 ? sell_skip_family(TypeId<T>::value, op_mult, op_add) : 0;
        bool                  bits   = family == 1;
        if (family == 2) {
            if (int rc = ax_stats<T>(g, st)) return rc;
            bits = min_plus_skippable<T>(g, init_bits);
        }
        const uint skip  = family == 2 ? init_bits : 0u;                          // the v value whose entries cannot change a fold
        const T    ident = family == 2 ? from_bits<T>(init_bits) : T(0);          // what a lane of a row-exclusive slice starts from
        uint*                 stats = g->d_vbits + 2 * (size_t) g->n_bit_words;
        uint*                 vbits = g->d_vbits + (size_t) g->bit_buffer * g->n_bit_words;
        uint*                 vnext = g->d_vbits + (size_t) (g->bit_buffer ^ 1u) * g->n_bit_words;
        if (bits) {
            g->bit_buffer ^= 1u;
            hipLaunchKernelGGL((sell_vprep_kernel<T>), dim3(SC::kPrepBlocks), dim3(256), 0, st, g->n_cols, static_cast<const T*>(v), g->d_rank_of, g->n_hot,
                               g->d_hot_cols, reinterpret_cast<T*>(g->d_hot_vals), vbits, g->d_vranked, stats, plan->n_rows, static_cast<const T*>(mask), op_select,
                               reinterpret_cast<unsigned long long*>(g->d_selbits), reinterpret_cast<uint4*>(vnext), g->n_bit_words / 4, static_cast<T*>(r),
                               from_bits<T>(init_bits), skip);
            hipLaunchKernelGGL((sell_mxv_dual_kernel<T, Mult, Add>), dim3(grid), dim3(SC::kTpb), 0, st, plan->n_rows, Ap, g->n_slices, g->d_vrow, g->d_vlen,
                               g->d_vslot, g->d_off, g->d_Aj2, g->d_Aj2r, g->x_uniform && !uniform_disabled() ? nullptr : g->d_Ax2,
                               reinterpret_cast<const T*>(g->d_hot_vals), g->n_hot,
                               static_cast<const T*>(mask), static_cast<const T*>(v), static_cast<T*>(r), reinterpret_cast<T*>(g->d_partial), Mult(op_mult),
                               Add(op_add), op_select, from_bits<T>(init_bits), stats, g->n_cols, mode == 2 ? 1 : 0, vbits, g->d_vranked, g->d_selbits, g->d_sorder, ident,
                               g->x_bits);
        } else {
            hipLaunchKernelGGL((sell_hot_pack_kernel<T>), dim3((g->n_hot + 255) / 256), dim3(256), 0, st, g->d_hot_cols, g->n_hot, static_cast<const T*>(v),
                               reinterpret_cast<T*>(g->d_hot_vals));
            hipLaunchKernelGGL((sell_mxv_kernel<T, Mult, Add>), dim3(grid), dim3(SC::kTpb), 0, st, plan->n_rows, Ap, g->n_slices, g->d_vrow, g->d_vlen, g->d_vslot,
                               g->d_off, g->d_Aj2, g->d_Ax2, reinterpret_cast<const T*>(g->d_hot_vals), g->n_hot, static_cast<const T*>(mask),
                               static_cast<const T*>(v), static_cast<T*>(r), reinterpret_cast<T*>(g->d_partial), Mult(op_mult), Add(op_add), op_select,
                               from_bits<T>(init_bits), g->d_sorder);
        }
        SPLAHIP_TRY(hipGetLastError());
        if (bits) {
            const uint cold_blocks = (g->n_cold + 255) / 256, comb_blocks = (g->n_lrows + 3) / 4;
            if (cold_blocks + comb_blocks) {
                hipLaunchKernelGGL((sell_post_kernel<T, Mult, Add>), dim3(cold_blocks + comb_blocks), dim3(256), 0, st, cold_blocks, g->n_cold, g->d_cold_rank,
                                   reinterpret_cast<const unsigned long long*>(g->d_cold_entry), stats, vbits, g->d_vranked, g->n_lrows, g->d_lr_row,
                                   g->d_lr_first, reinterpret_cast<const T*>(g->d_partial), static_cast<const T*>(mask), static_cast<T*>(r), Mult(op_mult),
                                   Add(op_add), op_select, from_bits<T>(init_bits));
                SPLAHIP_TRY(hipGetLastError());
            }
        } else if (g->n_lrows) {
            hipLaunchKernelGGL((sell_combine_kernel<T, Add>), dim3((g->n_lrows + 3) / 4), dim3(256), 0, st, g->n_lrows, g->d_lr_row, g->d_lr_first,
                               reinterpret_cast<const T*>(g->d_partial), static_cast<const T*>(mask), static_cast<T*>(r), Add(op_add), op_select,
                               from_bits<T>(init_bits));
            SPLAHIP_TRY(hipGetLastError());
        }
        return SPLAHIP_OK;
    }

#define SPLAHIP_SELL(MUL, ADD) launch_sell<T, StaBin<MUL>, StaBin<ADD>>(plan, Ap, mask, v, r, op_mult, op_add, op_select, init_bits, st)
    template<typename T>
    int sell_typed(const splahip_plan_s* plan, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                   uint32_t init_bits, hipStream_t st) {
        constexpr int type = TypeId<T>::value;
        if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS) return SPLAHIP_SELL(SPLAHIP_MULT, SPLAHIP_PLUS);
        if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN) return SPLAHIP_SELL(SPLAHIP_PLUS, SPLAHIP_MIN);
        if (op_mult == SPLAHIP_LAND && op_add == SPLAHIP_LOR) return SPLAHIP_SELL(SPLAHIP_LAND, SPLAHIP_LOR);
        if constexpr (type != SPLAHIP_FLOAT) {
            if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR) return SPLAHIP_SELL(SPLAHIP_BAND, SPLAHIP_BOR);
        }
        return launch_sell<T, DynBin, DynBin>(plan, Ap, mask, v, r, op_mult, op_add, op_select, init_bits, st);
    }
#undef SPLAHIP_SELL

}// namespace

namespace {
    template<typename T, class Mult, class Add>
    int launch_sell_vxm(const splahip_plan_s* plan, const uint32_t* Tp, const void* mask, const void* v, void* out, uint32_t* touched, int op_mult, int op_add,
                        int op_select, hipStream_t st) {
        const splahip_sell_s* g = plan->sell;
        hipLaunchKernelGGL((sell_hot_pack_kernel<T>), dim3((g->n_hot + 255) / 256), dim3(256), 0, st, g->d_hot_cols, g->n_hot, static_cast<const T*>(v),
                           reinterpret_cast<T*>(g->d_hot_vals));
        hipLaunchKernelGGL((sell_vxm_kernel<T, Mult, Add>), dim3(compute_units()), dim3(SC::kTpb), 0, st, plan->n_rows, Tp, g->n_slices, g->d_vrow, g->d_vlen,
                           g->d_vslot, g->d_off, g->d_Aj2, g->d_Ax2, reinterpret_cast<const T*>(g->d_hot_vals), g->n_hot, static_cast<const T*>(mask),
                           static_cast<const T*>(v), static_cast<T*>(out), touched, reinterpret_cast<T*>(g->d_partial), Mult(op_mult), Add(op_add), op_select,
                           g->d_sorder);
        SPLAHIP_TRY(hipGetLastError());
        if (g->n_lrows) {
            hipLaunchKernelGGL((sell_vxm_combine_kernel<T, Add>), dim3((g->n_lrows + 3) / 4), dim3(256), 0, st, g->n_lrows, g->d_lr_row, g->d_lr_first,
                               reinterpret_cast<const T*>(g->d_partial), static_cast<const T*>(mask), static_cast<T*>(out), touched, Add(op_add), op_select);
            SPLAHIP_TRY(hipGetLastError());
        }
        return SPLAHIP_OK;
    }
    template<typename T>
    int sell_vxm_typed(const splahip_plan_s* plan, const uint32_t* Tp, const void* mask, const void* v, void* out, uint32_t* touched, int op_mult, int op_add,
                       int op_select, hipStream_t st) {
        if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS)
            return launch_sell_vxm<T, StaBin<SPLAHIP_MULT>, StaBin<SPLAHIP_PLUS>>(plan, Tp, mask, v, out, touched, op_mult, op_add, op_select, st);
        if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN)
            return launch_sell_vxm<T, StaBin<SPLAHIP_PLUS>, StaBin<SPLAHIP_MIN>>(plan, Tp, mask, v, out, touched, op_mult, op_add, op_select, st);
        return launch_sell_vxm<T, DynBin, DynBin>(plan, Tp, mask, v, out, touched, op_mult, op_add, op_select, st);
    }
}// namespace

namespace splahip {

    int sell_skip_family(int type, int op_mult, int op_add) {
        if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN) return 2;// min-plus: every type
        if (type == SPLAHIP_FLOAT) return 0;                            // 0 * inf, -0 + 0: not bit-safe
        if (op_mult != SPLAHIP_BAND && op_mult != SPLAHIP_MULT && op_mult != SPLAHIP_LAND) return 0;
        if (op_add == SPLAHIP_BOR || op_add == SPLAHIP_PLUS || op_add == SPLAHIP_BXOR) return 1;
        return type == SPLAHIP_UINT && op_add == SPLAHIP_MAX ? 1 : 0;
    }

    void sell_release(splahip_sell_s* g) {
        if (!g) return;
        for (uint32_t* p : {g->d_vrow, g->d_vlen, g->d_vslot, g->d_off, g->d_Aj2, g->d_Ax2, g->d_hot_cols, g->d_hot_vals, g->d_lr_row, g->d_lr_first, g->d_partial,
                            g->d_rank_of, g->d_Aj2r, g->d_vbits, g->d_vranked, g->d_cold_rank, g->d_selbits, g->d_sorder})
            if (p) (void) hipFree(p);
        if (g->d_cold_entry) (void) hipFree(g->d_cold_entry);
        delete g;
    }

    int sell_vxm(const splahip_plan_s* plan_T, int type, const uint32_t* Tp, const void* mask, const void* v, void* out, uint32_t* touched, int op_mult,
                 int op_add, int op_select, void* stream) {
        hipStream_t st = as_stream(stream);
        switch (type) {
            case SPLAHIP_INT: return sell_vxm_typed<int32_t>(plan_T, Tp, mask, v, out, touched, op_mult, op_add, op_select, st);
            case SPLAHIP_UINT: return sell_vxm_typed<uint32_t>(plan_T, Tp, mask, v, out, touched, op_mult, op_add, op_select, st);
            case SPLAHIP_FLOAT: return sell_vxm_typed<float>(plan_T, Tp, mask, v, out, touched, op_mult, op_add, op_select, st);
        }
        return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }

    int sell_mxv(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                 uint32_t init_bits, void* stream) {
        hipStream_t st = as_stream(stream);
        switch (type) {
            case SPLAHIP_INT: return sell_typed<int32_t>(plan, Ap, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case SPLAHIP_UINT: return sell_typed<uint32_t>(plan, Ap, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case SPLAHIP_FLOAT: return sell_typed<float>(plan, Ap, mask, v, r, op_mult, op_add, op_select, init_bits, st);
        }
        return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }

}// namespace splahip

extern "C" int splahip_csr_plan_attach_sell(splahip_plan plan, uint32_t n_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, void* stream) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    SPLAHIP_REQUIRE(n_cols < SC::kHotFlag, "n_cols must be < 2^31");
    if (plan->sell) {
        sell_release(plan->sell);
        plan->sell = nullptr;
    }
    if (plan->nnz == 0 || plan->n_rows == 0 || n_cols == 0) return SPLAHIP_OK;// nothing to lay out: mxv keeps the CSR kernels
    SPLAHIP_REQUIRE(d_Ap && d_Aj && d_Ax, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t    s      = as_stream(stream);
    const uint32_t n_rows = plan->n_rows, nnz = plan->nnz;

    Scratch   tmp;
    auto*     g = new splahip_sell_s();
    g->n_cols   = n_cols;
    hipError_t e = hipSuccess;
    auto       bail = [&](const char* what) {
        sell_release(g);
        return fail(SPLAHIP_ERROR, "sliced-ELL plan failed (%s): %s", what, hipGetErrorString(e));
    };

    // SPLAHIP_SELL_TRACE=1: host wall time of the build's phases on stderr, each closed by a device synchronisation (diagnostic)
    static const bool trace = knob_env("SPLAHIP_SELL_TRACE") != nullptr;
    auto              t_last = std::chrono::steady_clock::now();
    auto              mark   = [&](const char* what) {
        if (!trace) return;
        (void) hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[sell attach] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_last).count());
        t_last = t1;
    };
    // 1. chunks per row, partial slots per long row, long-row index: three exclusive scans over n_rows + 1 counters
    uint32_t *vcount, *pcount, *lflag, *xcount, *vfirst, *pfirst, *lfirst, *xfirst;
    for (uint32_t** q : {&vcount, &pcount, &lflag, &xcount, &vfirst, &pfirst, &lfirst, &xfirst})
        if (e == hipSuccess) e = tmp.alloc(q, (size_t) n_rows + 1);
    if (e != hipSuccess) return bail("scratch");
    hipLaunchKernelGGL(sell_rowinfo_kernel, dim3((n_rows + 256) / 256), dim3(256), 0, s, d_Ap, n_rows, vcount, pcount, lflag, xcount);
    e = hipGetLastError();
    if (e == hipSuccess) e = exclusive_scan_u32(vcount, vfirst, (size_t) n_rows + 1, s);
    if (e == hipSuccess) e = exclusive_scan_u32(pcount, pfirst, (size_t) n_rows + 1, s);
    if (e == hipSuccess) e = exclusive_scan_u32(lflag, lfirst, (size_t) n_rows + 1, s);
    if (e == hipSuccess) e = exclusive_scan_u32(xcount, xfirst, (size_t) n_rows + 1, s);
    uint32_t n_chunks = 0;
    if (e == hipSuccess) e = hipMemcpy(&n_chunks, vfirst + n_rows, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&g->n_xslices, xfirst + n_rows, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&g->n_partials, pfirst + n_rows, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&g->n_lrows, lfirst + n_rows, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail("row scan");
    g->n_vrows  = g->n_xslices * kWave + (n_chunks + kWave - 1) / kWave * kWave;// row-exclusive slices first, then the sorted chunks
    g->n_slices = g->n_vrows / kWave;

    mark("1 row counts + scans");
    // 2. chunks in row order -> stable sort by descending length
    uint32_t *q_key, *q_id, *q_row, *q_len, *q_k0, *q_slot, *key_s, *order, *vk0;
    for (uint32_t** q : {&q_key, &q_id, &q_row, &q_len, &q_k0, &q_slot, &key_s, &order})
        if (e == hipSuccess) e = tmp.alloc(q, n_chunks);
    if (e == hipSuccess) e = tmp.alloc(&vk0, g->n_vrows);
    auto own = [&](uint32_t** q, size_t words) {
        if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(q), sizeof(uint32_t) * (words ? words : 1));
        if (e == hipSuccess) g->bytes += sizeof(uint32_t) * (words ? words : 1);
    };
    own(&g->d_vrow, g->n_vrows);
    own(&g->d_vlen, g->n_vrows);
    own(&g->d_vslot, g->n_vrows);
    own(&g->d_off, (size_t) g->n_slices + 1);
    own(&g->d_lr_row, g->n_lrows);
    own(&g->d_lr_first, (size_t) g->n_lrows + 1);
    own(&g->d_partial, g->n_partials);
    if (e != hipSuccess) return bail("chunk arrays");
    hipLaunchKernelGGL(sell_emit_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, d_Ap, n_rows, vfirst, pfirst, lfirst, q_key, q_id, q_row, q_len, q_k0, q_slot,
                       g->d_lr_row, g->d_lr_first, xfirst, g->d_vrow, g->d_vlen, vk0, g->d_vslot);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(g->d_lr_first + g->n_lrows, &g->n_partials, 4, hipMemcpyHostToDevice, s);
    unsigned key_bits = 1;
    while ((1u << key_bits) < SC::kC) ++key_bits;// keys are kC - len in [0, kC)
    if (e == hipSuccess) e = stable_sort_pairs(q_key, key_s, q_id, order, n_chunks, key_bits, s);
    if (e != hipSuccess) return bail("chunk sort");
    hipLaunchKernelGGL(sell_permute_kernel, dim3((g->n_vrows + 255) / 256), dim3(256), 0, s, n_chunks, g->n_xslices * kWave, g->n_vrows, order, q_row, q_len, q_k0,
                       q_slot, g->d_vrow, g->d_vlen, vk0, g->d_vslot);

    mark("2 chunk sort");
    // 3. slice offsets
    uint32_t* slots = nullptr;
    e               = tmp.alloc(&slots, (size_t) g->n_slices + 1);
    if (e != hipSuccess) return bail("slice scratch");
    // slice widths; the order the workgroups take the slices in: by descending width (stable)
    uint32_t *wkey = nullptr, *wkey_s = nullptr, *sid = nullptr;
    if (e == hipSuccess) e = tmp.alloc(&wkey, g->n_slices);
    if (e == hipSuccess) e = tmp.alloc(&wkey_s, g->n_slices);
    if (e == hipSuccess) e = tmp.alloc(&sid, g->n_slices);
    own(&g->d_sorder, g->n_slices);
    if (e != hipSuccess) return bail("slice order");
    hipLaunchKernelGGL(sell_width_kernel, dim3((g->n_slices + 256) / 256), dim3(256), 0, s, g->d_vlen, g->n_slices, slots, wkey, sid);
    e = hipGetLastError();
    if (e == hipSuccess) e = stable_sort_pairs(wkey, wkey_s, sid, g->d_sorder, g->n_slices, key_bits + 1, s);
    // the padded entry count must stay below 2^32 (offsets are 32-bit); padding is < 1 % after a global sort, so check the bound up front
    if (e == hipSuccess && (uint64_t) nnz + (uint64_t) g->n_slices * kWave * 2 >= 0xF0000000ull) {
        sell_release(g);
        return fail(SPLAHIP_NOT_IMPLEMENTED, "matrix too large for 32-bit slice offsets");
    }
    if (e == hipSuccess) e = exclusive_scan_u32(slots, g->d_off, (size_t) g->n_slices + 1, s);
    if (e == hipSuccess) e = hipMemcpy(&g->n_slots, g->d_off + g->n_slices, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail("slice scan");
    own(&g->d_Aj2, (size_t) g->n_slots + kTail);
    own(&g->d_Aj2r, (size_t) g->n_slots + kTail);
    own(&g->d_Ax2, (size_t) g->n_slots + kTail);

    mark("3 slice offsets");
    // 4. hot columns: the n_hot most referenced ones (stable: ties keep ascending column order)
    g->n_hot = n_cols < SC::kHot ? n_cols : SC::kHot;
    uint32_t *count, *ckey, *ccol, *ckey_s, *ccol_s;
    for (uint32_t** q : {&count, &ckey, &ccol, &ckey_s, &ccol_s})
        if (e == hipSuccess) e = tmp.alloc(q, n_cols);
    own(&g->d_hot_cols, g->n_hot);
    own(&g->d_hot_vals, (size_t) g->n_hot + 4);// whole 16-byte quads: the kernel copies the table with uint4 loads
    own(&g->d_rank_of, n_cols);
    own(&g->d_vranked, n_cols);
    // bitmap of v in rank order: at least the kHot words every workgroup copies to LDS, then the statistics (4 words per sell_vprep workgroup)
    g->n_bit_words = ((std::max<uint32_t>(n_cols / 32 + 1, SC::kHot) + 3u) & ~3u);// bits 1 .. n_cols (bit rank + 1)
    own(&g->d_vbits, 2 * (size_t) g->n_bit_words + 4 * SC::kPrepBlocks + 4);
    if (e == hipSuccess) e = hipMemsetAsync(g->d_vbits, 0, sizeof(uint32_t) * 2 * (size_t) g->n_bit_words, s);// both bitmaps start clear
    own(&g->d_selbits, 2 * ((size_t) n_rows / 64 + 8));// 64-bit words, whole 256-row wave steps
    if (e == hipSuccess) e = hipMemsetAsync(count, 0, sizeof(uint32_t) * (size_t) n_cols, s);
    if (e != hipSuccess) return bail("hot-column arrays");
    {
        const char*    se     = knob_env("SPLAHIP_SELL_COUNT_STRIDE");
        const uint32_t stride = nnz >= (1u << 22) ? std::max(1u, se ? (uint32_t) std::strtoul(se, nullptr, 10) : 8u) : 1u;
        const uint32_t n_samples = (nnz + stride - 1) / stride;
        hipLaunchKernelGGL(sell_col_count_kernel, dim3((n_samples + kCountSpan - 1) / kCountSpan), dim3(256), 0, s, d_Aj, nnz, stride, count);
    }
    hipLaunchKernelGGL(sell_col_key_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, s, count, n_cols, ckey, ccol);
    e = hipGetLastError();
    if (e == hipSuccess) e = stable_sort_pairs(ckey, ckey_s, ccol, ccol_s, n_cols, 32, s);
    if (e != hipSuccess) return bail("hot-column sort");
    hipLaunchKernelGGL(sell_rank_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, s, ccol_s, n_cols, g->n_hot, g->d_rank_of, g->d_hot_cols);

    mark("4 hot columns");
    // 4b. cold list of the sparse-vector form: entries of columns ranked beyond the LDS-resident bits, sorted by rank
    if (n_cols > SC::kBitsLds) {
        uint32_t *flag, *pos;
        if (e == hipSuccess) e = tmp.alloc(&flag, (size_t) nnz + 1);
        if (e == hipSuccess) e = tmp.alloc(&pos, (size_t) nnz + 1);
        if (e != hipSuccess) return bail("cold-list scratch");
        hipLaunchKernelGGL(sell_cold_flag_kernel, dim3((nnz + 256) / 256), dim3(256), 0, s, d_Aj, nnz, g->d_rank_of, flag);
        e = hipGetLastError();
        if (e == hipSuccess) e = exclusive_scan_u32(flag, pos, (size_t) nnz + 1, s);
        if (e == hipSuccess) e = hipMemcpy(&g->n_cold, pos + nnz, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail("cold-list scan");
        if (g->n_cold) {
            uint32_t*           rank_u = nullptr;
            unsigned long long *entry_u = nullptr, *entry_s = nullptr;
            if (e == hipSuccess) e = tmp.alloc(&rank_u, g->n_cold);
            if (e == hipSuccess) e = tmp.alloc(reinterpret_cast<uint32_t**>(&entry_u), 2 * (size_t) g->n_cold);
            own(&g->d_cold_rank, g->n_cold);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&entry_s), sizeof(unsigned long long) * g->n_cold);
            g->d_cold_entry = reinterpret_cast<uint64_t*>(entry_s);
            if (e != hipSuccess) return bail("cold-list arrays");
            hipLaunchKernelGGL(sell_cold_emit_kernel, dim3((nnz + 255) / 256), dim3(256), 0, s, d_Ap, n_rows, d_Aj, static_cast<const uint32_t*>(d_Ax), nnz,
                               g->d_rank_of, pos, rank_u, entry_u);
            size_t bytes = 0;
            e            = rocprim::radix_sort_pairs(nullptr, bytes, rank_u, g->d_cold_rank, entry_u, entry_s, g->n_cold, 0, 32, s);
            void* st_tmp = nullptr;
            if (e == hipSuccess) e = hipMalloc(&st_tmp, bytes ? bytes : 4);
            if (e == hipSuccess) e = rocprim::radix_sort_pairs(st_tmp, bytes, rank_u, g->d_cold_rank, entry_u, entry_s, g->n_cold, 0, 32, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (st_tmp) (void) hipFree(st_tmp);
            if (e != hipSuccess) return bail("cold-list sort");
        }
    }

    mark("4b cold list");
    // 5. the entries
    hipLaunchKernelGGL(sell_fill_kernel, dim3((g->n_slices + 3) / 4), dim3(256), 0, s, g->n_slices, g->n_xslices, g->d_off, g->d_vlen, vk0, d_Aj,
                       static_cast<const uint32_t*>(d_Ax), g->d_rank_of, g->n_hot, g->d_Aj2, g->d_Aj2r, g->d_Ax2);
    hipLaunchKernelGGL(sell_tail_kernel, dim3((kTail + 255) / 256), dim3(256), 0, s, g->n_slots, kTail, g->d_Aj2, g->d_Aj2r, g->d_Ax2);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail("fill");
    mark("5 entries");
    // 6. are all stored values one word?  (d_vbits is not in use yet: two of its words take the result)
    if (nnz > 0) {
        const uint range_init[2] = {0xFFFFFFFFu, 0u};
        uint       range[2]      = {0, 1};
        e = hipMemcpyAsync(g->d_vbits, range_init, 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(sell_ax_range_kernel, dim3(1024), dim3(256), 0, s, nnz, static_cast<const uint32_t*>(d_Ax), g->d_vbits);
            e = hipMemcpyAsync(range, g->d_vbits, 8, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipMemsetAsync(g->d_vbits, 0, 8, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail("value range");
        g->x_uniform = range[0] == range[1];
        g->x_bits    = range[0];
    }
    plan->sell = g;
    return SPLAHIP_OK;
}

// device memory held for one matrix besides its CSR arrays: the wave-tile plan, the sliced-ELL copy, the column-ordered-blocks copy (bytes; 0 = not built)
extern "C" int splahip_csr_plan_footprint(splahip_plan plan, uint64_t* plan_bytes, uint64_t* sell_bytes, uint64_t* cob_bytes) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    if (plan_bytes) *plan_bytes = (uint64_t) plan->n_items * 16 + 8 + (uint64_t) plan->ws_bytes;
    if (sell_bytes) *sell_bytes = plan->sell ? (uint64_t) plan->sell->bytes : 0;
    if (cob_bytes) *cob_bytes = plan->cob ? (uint64_t) plan->cob->bytes : 0;
    return SPLAHIP_OK;
}
// 1 when splahip_mxv_masked would run this semiring on the sliced-ELL copy even beside a column-ordered-blocks copy (the zero-family sparse-vector forms:
// a BFS step); a host that builds copies on demand needs the sliced-ELL one only for these
extern "C" int splahip_mxv_prefers_sell(int type, int op_mult, int op_add) { return splahip::sell_skip_family(type, op_mult, op_add) == 1 ? 1 : 0; }

extern "C" int splahip_csr_plan_sell_info(splahip_plan plan, uint32_t* n_slices, uint32_t* n_slots, uint32_t* n_hot, uint32_t* n_long_rows) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    const splahip_sell_s* g = plan->sell;
    if (n_slices) *n_slices = g ? g->n_slices : 0;
    if (n_slots) *n_slots = g ? g->n_slots : 0;
    if (n_hot) *n_hot = g ? g->n_hot : 0;
    if (n_long_rows) *n_long_rows = g ? g->n_lrows : 0;
    return SPLAHIP_OK;
}

extern "C" int splahip_csr_plan_sell_values(splahip_plan plan, int* uniform, uint32_t* bits) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    const splahip_sell_s* g = plan->sell;
    if (uniform) *uniform = g && g->x_uniform && !uniform_disabled() ? 1 : 0;
    if (bits) *bits = g ? g->x_bits : 0;
    return SPLAHIP_OK;
}

#ifdef SPLAHIP_SELL_TIMING
// DIAGNOSTIC BUILD ONLY: copy out the per-wave slice timings of the last sliced-ELL launch (12 x 4096 64-bit words)
extern "C" int splahip_debug_sell_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(splahip::g_sell_t), sizeof(unsigned long long) * 12 * 4096) == hipSuccess ? SPLAHIP_OK : SPLAHIP_ERROR;
}
#endif

