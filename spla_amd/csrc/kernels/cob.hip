// cob.hip — the "column-ordered blocks" copy of a device CSR matrix: built once per matrix on the device, kept with the plan
// (part of the AccCsr decoration, like the wave-tile items and the sliced-ELL copy), and the masked SpMV / pull-form vxm over it.
//
// Semantics served: reference Algo_mxv_masked_cpu, src/cpu/cpu_mxv.hpp:88-103, and the dense-frontier case of Algo_vxm_masked_cpu,
// src/cpu/cpu_vxm.hpp:92-125, for an op_add with an LDS atomic (mxv_cob.hpp: cob_acc_kind).  Kernels: mxv_cob.hpp.
//
// Build (all on the device; the host reads three counters):
//   1. cob_key_kernel     key(k) = ((range << cbits | column) << row_bits) | row_in_range for every stored entry, by one binary search of
//                         the entry's 2048-entry tile in Ap and one inside the tile's row span
//   2. a stable LSD radix sort of the keys on the (range, column) bits only, carrying the value word (rocprim, one-off per matrix):
//      inside a (range, column) run the entries keep ascending row order
//   3. cob_group_kernel   first sorted entry of every (range, column window) group by binary search; chunks per group; scan
//   4. cob_scatter_kernel one workgroup per chunk: pack cr / ax / hdr, pad a group's last chunk
//   5. cob_tile_kernels   tiles (<= tile_chunks consecutive chunks of one range), slab ids of the ranges that need several tiles
#include "block_utils.hpp"
#include "common.hpp"
#include "mxv_cob.hpp"
#include "ops.hpp"
#include "plan.hpp"

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace splahip;

namespace {

    constexpr uint32_t kKeyTile = 2048;// entries per workgroup of the key pass
    constexpr unsigned kTpb     = 1024;// threads of a tile's workgroup

    // first index in a[0..n) with a[index] > x
    __device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t x) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a[mid] <= x) lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    }

    __global__ __launch_bounds__(256) void cob_key_kernel(uint32_t n_rows, uint32_t nnz, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj,
                                                          uint32_t row_bits, uint32_t cbits, unsigned long long* __restrict__ key) {
        const uint32_t k0 = blockIdx.x * kKeyTile;
        const uint32_t k1 = min(nnz, k0 + kKeyTile);
        // rows spanned by this tile of the entry stream (uniform searches: every lane reads the same words)
        const uint32_t r_lo = upper_bound_u32(Ap, n_rows + 1, k0) - 1;
        const uint32_t r_hi = upper_bound_u32(Ap, n_rows + 1, k1 - 1) - 1;
        const uint32_t rmask = (1u << row_bits) - 1u;
        for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) {
            const uint32_t row = r_lo + upper_bound_u32(Ap + r_lo, r_hi - r_lo + 1, k) - 1;
            const unsigned long long range = row >> row_bits;
            key[k] = (((range << cbits) | Aj[k]) << row_bits) | (row & rmask);
        }
    }

    // start[g] = first sorted entry whose (range, window) group is >= g, g = 0 .. n_groups; chunks[g] = chunks of group g
    __global__ __launch_bounds__(256) void cob_group_kernel(uint32_t nnz, const unsigned long long* __restrict__ key, uint32_t shift, uint32_t n_groups,
                                                            uint32_t* __restrict__ start) {
        const uint32_t g = blockIdx.x * 256 + threadIdx.x;
        if (g > n_groups) return;
        uint32_t lo = 0, hi = nnz;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((key[mid] >> shift) < g) lo = mid + 1;
            else
                hi = mid;
        }
        start[g] = lo;
    }
    __global__ __launch_bounds__(256) void cob_chunk_count_kernel(uint32_t n_groups, const uint32_t* __restrict__ start, uint32_t* __restrict__ chunks) {
        const uint32_t g = blockIdx.x * 256 + threadIdx.x;
        if (g < n_groups) chunks[g] = (start[g + 1] - start[g] + kCobChunk - 1) / kCobChunk;
    }

    // one workgroup (256 threads = 256 slots) per chunk
    __global__ __launch_bounds__(256) void cob_scatter_kernel(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ val, const uint32_t* __restrict__ start,
                                                              const uint32_t* __restrict__ chunk_first, uint32_t n_groups, uint32_t row_bits, uint32_t col_bits,
                                                              uint32_t win_bits, uint32_t* __restrict__ cr, uint32_t* __restrict__ ax, uint32_t* __restrict__ hdr) {
        const uint32_t c = blockIdx.x;
        const uint32_t g = upper_bound_u32(chunk_first, n_groups + 1, c) - 1;// largest g with chunk_first[g] <= c (uniform)
        const uint32_t i = c - chunk_first[g];
        const uint32_t s0 = start[g], s1 = start[g + 1];
        // memory position p = 4 * lane + j holds the chunk's sorted entry 64 * j + lane: a lane's 16-byte load brings its four slots, and the
        // wave's j-th gather instruction covers 64 consecutive sorted entries (ascending, neighbouring columns)
        const uint32_t src = s0 + i * kCobChunk + 64 * (threadIdx.x & 3) + (threadIdx.x >> 2);
        uint32_t       w = (1u << row_bits) << col_bits, x = 0;// padding: row field R (the spare LDS word), column offset 0
        if (src < s1) {
            const unsigned long long k = key[src];
            const uint32_t           rl = (uint32_t) k & ((1u << row_bits) - 1u);
            const uint32_t           co = (uint32_t) (k >> row_bits) & ((1u << col_bits) - 1u);
            w                           = (rl << col_bits) | co;
            x                           = val[src];
        }
        cr[(size_t) c * kCobChunk + threadIdx.x] = w;
        ax[(size_t) c * kCobChunk + threadIdx.x] = x;
        if (threadIdx.x == 0) {
            const uint32_t window = win_bits ? (g & ((1u << win_bits) - 1u)) : 0u;
            hdr[c]                = window << col_bits;
        }
    }

    __global__ __launch_bounds__(256) void cob_padchunk_kernel(uint32_t c, uint32_t row_bits, uint32_t col_bits, uint32_t* __restrict__ cr, uint32_t* __restrict__ ax,
                                                               uint32_t* __restrict__ hdr) {
        cr[(size_t) c * kCobChunk + threadIdx.x] = (1u << row_bits) << col_bits;
        ax[(size_t) c * kCobChunk + threadIdx.x] = 0;
        if (threadIdx.x == 0) hdr[c] = 0;
    }

    // smallest and largest value word of the matrix (equal <=> one value, as in an adjacency matrix of ones)
    __global__ __launch_bounds__(256) void cob_ax_range_kernel(uint32_t n, const uint32_t* __restrict__ x, uint32_t* __restrict__ out) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
            const uint32_t w = x[k];
            lo = min(lo, w), hi = max(hi, w);
        }
        lo = ~wave_max(~lo);
        hi = wave_max(hi);
        if (lane_id() == 0) {
            atomicMin(&out[0], lo);
            atomicMax(&out[1], hi);
        }
    }
    // per call, sparse-input-vector form: bit c of the bitmap = v[c] is not the value that cannot change a fold (0 for the zero family,
    // init — the step's "infinity" — for min-plus); a wave's ballot is two bitmap words
    template<typename T>
    __global__ __launch_bounds__(256) void cob_vbits_kernel(uint32_t n, const T* __restrict__ v, uint32_t* __restrict__ bits, uint32_t skip) {
        const uint32_t           c  = blockIdx.x * 256 + threadIdx.x;
        const unsigned long long b  = __ballot(c < n && to_bits(v[c]) != skip);
        if (lane_id() == 0) {
            bits[c >> 5]       = (uint32_t) b;
            bits[(c >> 5) + 1] = (uint32_t) (b >> 32);
        }
    }

    // rows of ranges that hold no entry at all: r = init (grid.y = empty range)
    template<typename T>
    __global__ __launch_bounds__(256) void cob_fill_kernel(const uint32_t* __restrict__ empty_ranges, uint32_t n_rows, uint32_t row_bits, T* __restrict__ r, T init) {
        const uint32_t i   = blockIdx.x * 256 + threadIdx.x;
        const uint32_t row = (empty_ranges[blockIdx.y] << row_bits) + i;
        if (i < (1u << row_bits) && row < n_rows) r[row] = init;
    }

    struct Scratch {
        std::vector<void*> ptrs;
        ~Scratch() {
            for (void* p : ptrs)
                if (p) (void) hipFree(p);
        }
        template<typename U>
        hipError_t alloc(U** out, size_t count) {
            void*      p = nullptr;
            hipError_t e = device_malloc(&p, sizeof(U) * (count ? count : 1));
            if (e == hipSuccess) ptrs.push_back(p);
            *out = static_cast<U*>(p);
            return e;
        }
    };

    uint32_t bits_for(uint64_t count) {// bits needed to tell `count` values apart (0 for count <= 1)
        uint32_t b = 0;
        while ((uint64_t(1) << b) < count) ++b;
        return b;
    }

    bool cob_disabled() {
        const char* e = std::getenv("SPLAHIP_COB");
        return e && e[0] == '0';
    }
    uint32_t compute_units() {
        int             dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) return (uint32_t) prop.multiProcessorCount;
        return 256;
    }
    uint32_t env_u32(const char* name, uint32_t dflt) {
        const char* e = std::getenv(name);
        return e && *e ? (uint32_t) std::strtoul(e, nullptr, 10) : dflt;
    }

    // set by cob_mxv_accumulate around its call (per host thread): the launch below then folds onto the values r already holds
    thread_local uint32_t tl_accumulate = 0;

    static thread_local uint32_t tl_skip_bits = 0;// the v value whose entries the sparse-vector form skips in the running call (0; init for min-plus)

    // any negative / NaN value, and the largest value, of the copy's value words (padding slots hold 0)
    template<typename T>
    __global__ __launch_bounds__(256) void cob_ax_stats_kernel(size_t n, const T* __restrict__ x, uint32_t* __restrict__ out) {
        uint32_t bad = 0, mx = 0;
        for (size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t) gridDim.x * 256) {
            const T a = x[k];
            if constexpr (std::is_same<T, uint32_t>::value) mx = max(mx, a);
            else {
                if (a < T(0) || a != a) bad = 1;
                else if constexpr (std::is_same<T, int32_t>::value)
                    mx = max(mx, (uint32_t) a);
            }
        }
        bad = wave_max(bad);
        mx  = wave_max(mx);
        if (lane_id() == 0) {
            if (bad) atomicOr(&out[0], 1u);
            atomicMax(&out[1], mx);
        }
    }
    template<typename T>
    int cob_ax_stats(const splahip_cob_s* g, hipStream_t st) {
        if (g->ax_type == TypeId<T>::value) return SPLAHIP_OK;
        DevMem out;
        SPLAHIP_TRY(out.alloc(st, 8));
        SPLAHIP_TRY(hipMemsetAsync(out.p, 0, 8, st));
        hipLaunchKernelGGL((cob_ax_stats_kernel<T>), dim3(1024), dim3(256), 0, st, (size_t) g->n_chunks * kCobChunk, reinterpret_cast<const T*>(g->d_ax), out.as<uint32_t>());
        uint32_t h[2] = {1, 0};
        SPLAHIP_TRY(hipMemcpyAsync(h, out.p, 8, hipMemcpyDeviceToHost, st));
        SPLAHIP_TRY(hipStreamSynchronize(st));
        g->ax_bad = h[0], g->ax_max = h[1], g->ax_type = TypeId<T>::value;
        return SPLAHIP_OK;
    }

    template<typename T, class Mult, int ACC, bool VXM, int MODE = 0>
    int launch_cob(const splahip_cob_s* g, const uint32_t* Ap, const void* mask, const void* v, void* r, uint32_t* touched, int op_mult, int op_add, int op_select,
                   uint32_t init_bits, hipStream_t st) {
        constexpr int               W  = sizeof(typename CobSlice<T, ACC>::word) == 8 ? 1 : 0;
        const splahip_cob_sched_s&  sc = g->sched[W];
        CobArgs                     a;
        a.span_first = sc.d_span_first;
        a.tiles      = reinterpret_cast<const CobTile*>(sc.d_tiles);
        a.hdr      = g->d_hdr;
        a.cr       = g->d_cr;
        a.ax       = g->d_ax;
        a.Ap       = Ap;
        a.slabs    = g->d_slabs;
        a.touched  = touched;
        a.n_rows   = g->n_rows;
        a.row_bits = g->row_bits;
        a.col_bits = g->col_bits;
        a.pad_chunk = g->n_chunks;
        a.accumulate = VXM ? 0u : tl_accumulate;
        a.vbits      = g->d_vbits;
        a.x_uniform  = g->x_bits;
        if constexpr (MODE != 0) {// the bitmap of v for this call
            hipLaunchKernelGGL((cob_vbits_kernel<T>), dim3((g->n_cols + 255) / 256), dim3(256), 0, st, g->n_cols, static_cast<const T*>(v), g->d_vbits, tl_skip_bits);
            SPLAHIP_TRY(hipGetLastError());
        }
        const T        init = from_bits<T>(init_bits);
        const uint32_t R    = 1u << g->row_bits;
        if (sc.n_tiles) {
            static bool lds_raised = false;// a range of 16 Ki rows plus the spare words is just over the 64 KB a kernel gets without asking
            if (!lds_raised) {
                SPLAHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&cob_mxv_kernel<T, Mult, DynBin, ACC, VXM, kTpb, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                160 * 1024));
                lds_raised = true;
            }
            hipLaunchKernelGGL((cob_mxv_kernel<T, Mult, DynBin, ACC, VXM, kTpb, MODE>), dim3(sc.n_spans), dim3(kTpb), (R + 4) * sizeof(typename CobSlice<T, ACC>::word), st, a, static_cast<const T*>(v),
                               static_cast<const T*>(mask), static_cast<T*>(r), Mult(op_mult), DynBin(op_add), op_select, init);
            SPLAHIP_TRY(hipGetLastError());
        }
        if (sc.n_multi) {
            const uint32_t bx = (R / 4 + 255) / 256;
            hipLaunchKernelGGL((cob_combine_kernel<T, DynBin, ACC, VXM, 256>), dim3(bx, sc.n_multi), dim3(256), 0, st, reinterpret_cast<const CobMulti*>(sc.d_multi), g->d_slabs, Ap,
                               static_cast<const T*>(mask), static_cast<T*>(r), touched, g->n_rows, g->row_bits, DynBin(op_add), op_select, init, a.accumulate);
            SPLAHIP_TRY(hipGetLastError());
        }
        if (g->n_empty && !a.accumulate) {// ranges without an entry: r = init (mxv) / touched = 0 (vxm); an accumulating pass leaves them alone
            if constexpr (VXM) hipLaunchKernelGGL((cob_fill_kernel<uint32_t>), dim3((R + 255) / 256, g->n_empty), dim3(256), 0, st, g->d_empty, g->n_rows, g->row_bits, touched, 0u);
            else
                hipLaunchKernelGGL((cob_fill_kernel<T>), dim3((R + 255) / 256, g->n_empty), dim3(256), 0, st, g->d_empty, g->n_rows, g->row_bits, static_cast<T*>(r), init);
            SPLAHIP_TRY(hipGetLastError());
        }
        return SPLAHIP_OK;
    }

    template<typename T, int ACC, bool VXM>
    int cob_by_mult(const splahip_cob_s* g, const uint32_t* Ap, const void* mask, const void* v, void* r, uint32_t* touched, int op_mult, int op_add, int op_select,
                    uint32_t init_bits, hipStream_t st) {
        // the sparse-input-vector form (mxv_cob.hpp MODE 1 / 2) for the semirings in which a zero of v cannot change a row (sell_skip_family == 1): int / uint only
        if constexpr (!VXM && !std::is_same<T, float>::value && (ACC == COB_OR || ACC == COB_ADD || ACC == COB_XOR || ACC == COB_MAX)) {
            const char* be = std::getenv("SPLAHIP_COB_BITS");
            if (g->d_vbits && !(be && be[0] == '0') && !tl_accumulate && sell_skip_family(TypeId<T>::value, op_mult, op_add) == 1) {
                const char* ue = std::getenv("SPLAHIP_COB_UNIFORM");
                if (g->x_uniform && !(ue && ue[0] == '0')) {
                    if (op_mult == SPLAHIP_BAND) return launch_cob<T, StaBin<SPLAHIP_BAND>, ACC, VXM, 2>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                    return launch_cob<T, DynBin, ACC, VXM, 2>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                }
                if (op_mult == SPLAHIP_BAND) return launch_cob<T, StaBin<SPLAHIP_BAND>, ACC, VXM, 1>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                return launch_cob<T, DynBin, ACC, VXM, 1>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            }
        }
        // the same form for min-plus (PLUS / MIN on int / uint, sell_skip_family == 2: the shortest-path step): an entry whose v value is `init` cannot
        // lower a minimum that starts at init, provided no matrix value is negative and init + max(A) does not wrap (checked once per type / per call)
        if constexpr (!VXM && !std::is_same<T, float>::value && ACC == COB_MIN) {
            const char* be = std::getenv("SPLAHIP_COB_BITS");
            // (worth it once v no longer fits the L2s: R-MAT 24, 67 MB of v: 0.83 -> 0.70 ms; R-MAT 22, 16.8 MB: 178 -> 204 us.  SPLAHIP_COB_BITS=2 forces it)
            const bool big = g->n_cols >= (1u << 23) || (be && be[0] == '2');
            if (g->d_vbits && big && !(be && be[0] == '0') && !tl_accumulate && sell_skip_family(TypeId<T>::value, op_mult, op_add) == 2) {
                if (int rc = cob_ax_stats<T>(g, st)) return rc;
                const bool fits = std::is_same<T, int32_t>::value ? (long long) (int32_t) init_bits + (long long) g->ax_max <= 0x7FFFFFFFll
                                                                  : (unsigned long long) init_bits + (unsigned long long) g->ax_max <= 0xFFFFFFFFull;
                if (!g->ax_bad && fits) {
                    tl_skip_bits = init_bits;
                    const int rc = launch_cob<T, StaBin<SPLAHIP_PLUS>, ACC, VXM, 1>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                    tl_skip_bits = 0;
                    return rc;
                }
            }
        }
        // the semirings of the configs get a compile-time op_mult; every other pair runs the runtime-switched functor
        if constexpr (ACC == COB_ADD) {
            if (op_mult == SPLAHIP_MULT) return launch_cob<T, StaBin<SPLAHIP_MULT>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        if constexpr (ACC == COB_MIN) {
            if (op_mult == SPLAHIP_PLUS) return launch_cob<T, StaBin<SPLAHIP_PLUS>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        if constexpr (ACC == COB_OR && !std::is_same<T, float>::value) {
            if (op_mult == SPLAHIP_BAND) return launch_cob<T, StaBin<SPLAHIP_BAND>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        if constexpr (ACC == COB_LOR) {
            if (op_mult == SPLAHIP_LAND) return launch_cob<T, StaBin<SPLAHIP_LAND>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        return launch_cob<T, DynBin, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
    }

    template<typename T, bool VXM>
    int cob_typed(const splahip_cob_s* g, const uint32_t* Ap, const void* mask, const void* v, void* r, uint32_t* touched, int op_mult, int op_add, int op_select,
                  uint32_t init_bits, hipStream_t st) {
        constexpr int type = TypeId<T>::value;
        switch (cob_acc_kind(op_add, type)) {
            case COB_ADD: return cob_by_mult<T, COB_ADD, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            case COB_LOR: return cob_by_mult<T, COB_LOR, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            case COB_LAND: return cob_by_mult<T, COB_LAND, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            default: break;
        }
        if constexpr (!std::is_same<T, float>::value) {
            switch (cob_acc_kind(op_add, type)) {
                case COB_MIN: return cob_by_mult<T, COB_MIN, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_MAX: return cob_by_mult<T, COB_MAX, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_OR: return cob_by_mult<T, COB_OR, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_AND: return cob_by_mult<T, COB_AND, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_XOR: return cob_by_mult<T, COB_XOR, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                default: break;
            }
        }
        return fail(SPLAHIP_NOT_IMPLEMENTED, "the column-ordered-blocks form has no LDS atomic for this op_add");
    }

}// namespace

namespace splahip {

    void cob_release(splahip_cob_s* g) {
        if (!g) return;
        for (void* p : {(void*) g->d_hdr, (void*) g->d_cr, (void*) g->d_ax, (void*) g->d_slabs, (void*) g->d_empty, (void*) g->d_vbits, (void*) g->sched[0].d_span_first,
                        (void*) g->sched[0].d_tiles, (void*) g->sched[0].d_multi, (void*) g->sched[1].d_span_first, (void*) g->sched[1].d_tiles, (void*) g->sched[1].d_multi})
            if (p) (void) hipFree(p);
        delete g;
    }

    bool cob_serves(const splahip_plan_s* plan, int type, int op_add, uint32_t n_cols) {
        if (!plan || !plan->cob || plan->cob->n_cols != n_cols || cob_disabled()) return false;
        const int acc = cob_acc_kind(op_add, type);
        if (acc == COB_NONE) return false;
        const bool wide = acc == COB_ADD && type == SPLAHIP_FLOAT;// fp32 sums keep a double slice
        return plan->cob->sched[wide ? 1 : 0].n_spans != 0;
    }

    int cob_mxv(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                uint32_t init_bits, void* stream) {
        hipStream_t st = as_stream(stream);
        switch (type) {
            case SPLAHIP_INT: return cob_typed<int32_t, false>(plan->cob, Ap, mask, v, r, nullptr, op_mult, op_add, op_select, init_bits, st);
            case SPLAHIP_UINT: return cob_typed<uint32_t, false>(plan->cob, Ap, mask, v, r, nullptr, op_mult, op_add, op_select, init_bits, st);
            case SPLAHIP_FLOAT: return cob_typed<float, false>(plan->cob, Ap, mask, v, r, nullptr, op_mult, op_add, op_select, init_bits, st);
        }
        return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }

    // second column part of a row block (dist.hip): r[row] = add(r[row], fold of this part's products) for the selected rows that hold an entry here
    int cob_mxv_accumulate(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add,
                           int op_select, void* stream) {
        tl_accumulate = 1;
        const int st  = cob_mxv(plan, type, Ap, mask, v, r, op_mult, op_add, op_select, 0u, stream);
        tl_accumulate = 0;
        return st;
    }

    // the pull form of vxm with every index of v stored, over the copy attached to the TRANSPOSED matrix's plan: out values and touched flags per
    // output index.  Only an op_add with a true identity (not LOR / LAND: a lone product must come out as itself, not as its truth value)
    bool cob_vxm_serves(const splahip_plan_s* plan_T, int type, int op_add, uint32_t n_in) {
        return op_add != SPLAHIP_LOR && op_add != SPLAHIP_LAND && cob_serves(plan_T, type, op_add, n_in);
    }
    int cob_vxm(const splahip_plan_s* plan_T, int type, const uint32_t* Tp, const void* mask, const void* v, void* out, uint32_t* touched, int op_mult, int op_add,
                int op_select, void* stream) {
        hipStream_t st = as_stream(stream);
        switch (type) {
            case SPLAHIP_INT: return cob_typed<int32_t, true>(plan_T->cob, Tp, mask, v, out, touched, op_mult, op_add, op_select, 0u, st);
            case SPLAHIP_UINT: return cob_typed<uint32_t, true>(plan_T->cob, Tp, mask, v, out, touched, op_mult, op_add, op_select, 0u, st);
            case SPLAHIP_FLOAT: return cob_typed<float, true>(plan_T->cob, Tp, mask, v, out, touched, op_mult, op_add, op_select, 0u, st);
        }
        return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }

}// namespace splahip

extern "C" int splahip_csr_plan_attach_cob(splahip_plan plan, uint32_t n_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, uint32_t row_bits,
                                           uint32_t tile_chunks, void* stream) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    if (plan->cob) {
        cob_release(plan->cob);
        plan->cob = nullptr;
    }
    if (plan->nnz == 0 || plan->n_rows == 0 || n_cols == 0) return SPLAHIP_OK;// nothing to lay out: mxv keeps the CSR kernels
    SPLAHIP_REQUIRE(d_Ap && d_Aj && d_Ax, "null device pointer");
    if (row_bits == 0) row_bits = env_u32("SPLAHIP_COB_ROW_BITS", 14);
    if (tile_chunks == 0) tile_chunks = env_u32("SPLAHIP_COB_TILE_CHUNKS", 0);// 0: pieces are cut at range and span boundaries only
    SPLAHIP_REQUIRE(row_bits >= 8 && row_bits <= 15, "row_bits must be in [8, 15] (the range's result slice lives in LDS: 4 << row_bits bytes)");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t    s      = as_stream(stream);
    const uint32_t n_rows = plan->n_rows, nnz = plan->nnz;

    const uint32_t col_bits = 31 - row_bits;                              // the row field is one bit wider than a row needs (padding slots: row value R)
    const uint32_t cbits    = std::max(bits_for(n_cols), col_bits);      // column field of the sort key: at least one whole window
    const uint32_t win_bits = cbits - col_bits;                          // column windows per range = 2^win_bits
    const uint32_t n_ranges = (uint32_t) (((uint64_t) n_rows + (1u << row_bits) - 1) >> row_bits);
    const uint64_t n_groups64 = (uint64_t) n_ranges << win_bits;
    // every (range, window) group pads its last chunk: with more groups than nnz / 512 the copy would be mostly padding (a hypersparse
    // matrix) — the CSR kernels keep such a matrix
    if (n_groups64 > (uint64_t) nnz / 512 + 64 || n_groups64 >= (1u << 30)) return SPLAHIP_OK;
    const uint32_t n_groups = (uint32_t) n_groups64;
    SPLAHIP_REQUIRE(row_bits + cbits + bits_for(n_ranges) <= 64, "key does not fit 64 bits");

    Scratch tmp;
    auto*   g   = new splahip_cob_s();
    g->n_rows   = n_rows;
    g->n_cols   = n_cols;
    g->row_bits = row_bits;
    g->col_bits = col_bits;
    hipError_t e    = hipSuccess;
    auto       bail = [&](const char* what) {
        cob_release(g);
        return fail(SPLAHIP_ERROR, "column-ordered-blocks plan failed (%s): %s", what, hipGetErrorString(e));
    };

    // 1 + 2: keys, stable sort on the (range, column) bits
    unsigned long long *key = nullptr, *key_s = nullptr;
    uint32_t*           val_s = nullptr;
    e                         = tmp.alloc(&key, nnz);
    if (e == hipSuccess) e = tmp.alloc(&key_s, nnz);
    if (e == hipSuccess) e = tmp.alloc(&val_s, nnz);
    if (e != hipSuccess) return bail("scratch");
    hipLaunchKernelGGL(cob_key_kernel, dim3((nnz + kKeyTile - 1) / kKeyTile), dim3(256), 0, s, n_rows, nnz, d_Ap, d_Aj, row_bits, cbits, key);
    e = hipGetLastError();
    {
        size_t      bytes = 0;
        const auto* vin   = static_cast<const uint32_t*>(d_Ax);
        const unsigned b0 = row_bits, b1 = row_bits + cbits + bits_for(n_ranges);
        if (b1 == b0) {// one range, one column: the entries are in order already
            if (e == hipSuccess) e = hipMemcpyAsync(key_s, key, sizeof(unsigned long long) * nnz, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(val_s, vin, sizeof(uint32_t) * nnz, hipMemcpyDeviceToDevice, s);
        } else {
            if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, bytes, key, key_s, vin, val_s, (size_t) nnz, b0, b1, s);
            void* st_tmp = nullptr;
            if (e == hipSuccess) e = device_malloc(&st_tmp, bytes ? bytes : 4);
            if (e == hipSuccess) {
                tmp.ptrs.push_back(st_tmp);
                e = rocprim::radix_sort_pairs(st_tmp, bytes, key, key_s, vin, val_s, (size_t) nnz, b0, b1, s);
            }
        }
    }
    if (e != hipSuccess) return bail("sort");

    // 3: groups -> chunks
    uint32_t *start = nullptr, *chunk_first = nullptr;
    e = tmp.alloc(&start, (size_t) n_groups + 1);
    if (e == hipSuccess) e = tmp.alloc(&chunk_first, (size_t) n_groups + 1);
    if (e != hipSuccess) return bail("scratch");
    hipLaunchKernelGGL(cob_group_kernel, dim3((n_groups + 256) / 256), dim3(256), 0, s, nnz, key_s, row_bits + col_bits, n_groups, start);
    hipLaunchKernelGGL(cob_chunk_count_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, s, n_groups, start, chunk_first);
    hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk_first, n_groups);
    e = hipGetLastError();
    uint32_t n_chunks = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n_chunks, chunk_first + n_groups, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail("groups");
    g->n_chunks = n_chunks;

    // 4: the packed streams
    auto own = [&](auto** q, size_t count) {
        using U = std::remove_pointer_t<std::remove_pointer_t<decltype(q)>>;
        if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(q), sizeof(U) * (count ? count : 1));
    };
    own(&g->d_cr, ((size_t) n_chunks + 1) * kCobChunk);// + one all-padding chunk (the main loop's steps beyond a tile's end walk it)
    own(&g->d_ax, ((size_t) n_chunks + 1) * kCobChunk);
    own(&g->d_hdr, (size_t) n_chunks + 1);
    if (e != hipSuccess) return bail("hipMalloc");
    hipLaunchKernelGGL(cob_padchunk_kernel, dim3(1), dim3(256), 0, s, n_chunks, row_bits, col_bits, g->d_cr, g->d_ax, g->d_hdr);
    hipLaunchKernelGGL(cob_scatter_kernel, dim3(n_chunks), dim3(256), 0, s, key_s, val_s, start, chunk_first, n_groups, row_bits, col_bits, win_bits, g->d_cr, g->d_ax,
                       g->d_hdr);
    e = hipGetLastError();
    if (e != hipSuccess) return bail("scatter");

    // 5: the work schedules, on the host (a few hundred pieces).  A piece = consecutive chunks of ONE range (the LDS slice belongs to a range) =
    // one workgroup.  A range of at most `share` chunks — what one workgroup slot of the chip would get if the chunks were dealt evenly — is a
    // single piece and finishes its rows itself; a larger range is cut into equal pieces of at most `share` chunks that store partial slabs
    // (cob_combine_kernel folds them in piece order).  Workgroups are launched largest piece first, so the small pieces fill the tail.
    // tile_chunks > 0 caps a piece explicitly (tests: many pieces and slabs on small inputs).
    std::vector<uint32_t> h_first((size_t) n_groups + 1);
    e = hipMemcpyAsync(h_first.data(), chunk_first, sizeof(uint32_t) * h_first.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail("schedule");
    std::vector<uint32_t> range_chunk((size_t) n_ranges + 1);
    for (uint32_t q = 0; q <= n_ranges; ++q) range_chunk[q] = h_first[(size_t) q << win_bits];
    std::vector<uint32_t> empty;
    for (uint32_t q = 0; q < n_ranges; ++q)
        if (range_chunk[q + 1] == range_chunk[q]) empty.push_back(q);
    g->n_empty = (uint32_t) empty.size();
    own(&g->d_empty, empty.size());
    if (e == hipSuccess && !empty.empty()) e = hipMemcpyAsync(g->d_empty, empty.data(), 4 * empty.size(), hipMemcpyHostToDevice, s);
    const uint32_t cus = compute_units();
    uint32_t       max_slabs = 0;
    for (int w = 0; w < 2 && e == hipSuccess; ++w) {
        splahip_cob_sched_s& sc     = g->sched[w];
        const size_t         lds    = ((size_t) (1u << row_bits) + 4) * (w ? 8 : 4);
        const uint32_t       per_cu = (uint32_t) std::min<size_t>(env_u32("SPLAHIP_COB_PER_CU", 2), (160 * 1024) / lds);// 1024-thread workgroups: at most 2 per CU
        if (per_cu == 0) continue;                                                         // this slice width does not fit: not served
        const uint32_t slots = cus * per_cu;
        // piece size in % of a slot's even share.  Larger pieces mean fewer slabs to write and merge, smaller ones a better-balanced tail; measured
        // (fp32, one workgroup per CU, pieces in brackets): R-MAT 22 — 100 / 140 / 200 / 300 % = 171.9 (402) / 171.1 (378) / 188 (329) / 271 us (280);
        // R-MAT 23 — 405.8 (657) / 373.8 (598) / 380.5 (584) / 494 us (536).  SPLAHIP_COB_SHARE_PCT overrides.
        auto pieces_at = [&](uint32_t pct) -> uint32_t {
            const uint32_t sh = std::max(1u, (n_chunks + slots - 1) / slots * pct / 100);
            uint32_t       n  = 0;
            for (uint32_t q = 0; q < n_ranges; ++q) n += (range_chunk[q + 1] - range_chunk[q] + sh - 1) / sh;
            return n;
        };
        // (the 4-byte schedule, two workgroups per CU, keeps 100 %: the SSSP step at R-MAT 24 runs 0.70 ms with 100 % and 0.76 ms with 140 %)
        const uint32_t share_pct = env_u32("SPLAHIP_COB_SHARE_PCT", w ? 140 : 100);
        const uint32_t share     = std::max(1u, (n_chunks + slots - 1) / slots * share_pct / 100);
        if (std::getenv("SPLAHIP_COB_TRACE")) std::fprintf(stderr, "[cob] schedule %d: slots %u, piece size %u %% of a share = %u chunks, %u pieces\n", w, slots, share_pct, share, pieces_at(share_pct));
        const uint32_t cap   = tile_chunks ? tile_chunks : share;
        std::vector<uint32_t> tiles, multi;
        std::vector<uint32_t> range_pieces(n_ranges, 0u);
        for (uint32_t q = 0; q < n_ranges; ++q) {
            const uint32_t cc = range_chunk[q + 1] - range_chunk[q];
            if (cc == 0) continue;
            const uint32_t np = (cc + cap - 1) / cap, per = cc / np, rem = cc % np;
            uint32_t       f  = range_chunk[q];
            for (uint32_t t = 0; t < np; ++t) {
                const uint32_t len = per + (t < rem ? 1u : 0u);
                tiles.insert(tiles.end(), {f, len, q, kCobNoSlab});
                f += len;
            }
            range_pieces[q] = np;
        }
        // slab ids in piece order: the pieces of a range are consecutive
        uint32_t n_slabs = 0;
        for (size_t t = 0; t < tiles.size() / 4; ++t) {
            const uint32_t rq = tiles[4 * t + 2];
            if (range_pieces[rq] > 1) {
                if (multi.empty() || multi[multi.size() - 4] != rq) multi.insert(multi.end(), {rq, n_slabs, 0u, 0u});
                ++multi[multi.size() - 2];
                tiles[4 * t + 3] = n_slabs++;
            }
        }
        // launch order: largest piece first (stable: equal pieces keep range order)
        const uint32_t        n_tiles = (uint32_t) (tiles.size() / 4);
        std::vector<uint32_t> order(n_tiles), sorted(tiles.size()), span_first((size_t) n_tiles + 1);
        for (uint32_t t = 0; t < n_tiles; ++t) order[t] = t;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return tiles[4 * x + 1] > tiles[4 * y + 1]; });
        for (uint32_t t = 0; t < n_tiles; ++t)
            for (int k = 0; k < 4; ++k) sorted[4 * t + k] = tiles[4 * order[t] + k];
        for (uint32_t t = 0; t <= n_tiles; ++t) span_first[t] = t;
        sc.n_spans = n_tiles;
        sc.n_tiles = n_tiles;
        sc.n_slabs = n_slabs;
        sc.n_multi = (uint32_t) (multi.size() / 4);
        max_slabs  = std::max(max_slabs, n_slabs);
        own(&sc.d_span_first, span_first.size());
        own(&sc.d_tiles, sorted.size());
        own(&sc.d_multi, multi.size());
        if (e == hipSuccess) e = hipMemcpyAsync(sc.d_span_first, span_first.data(), 4 * span_first.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess && !sorted.empty()) e = hipMemcpyAsync(sc.d_tiles, sorted.data(), 4 * sorted.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess && !multi.empty()) e = hipMemcpyAsync(sc.d_multi, multi.data(), 4 * multi.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);// the vectors go out of scope
    }
    own(&g->d_slabs, (size_t) max_slabs << row_bits);
    own(&g->d_vbits, ((size_t) n_cols + 255) / 256 * 8 + 2);// every wave of cob_vbits_kernel writes its two words
    if (e != hipSuccess) return bail("schedule upload");
    {// one value word in the whole matrix?  (the sparse-vector form then skips the value stream)
        uint32_t* rng = nullptr;
        e             = tmp.alloc(&rng, 2);
        const uint32_t init_rng[2] = {0xFFFFFFFFu, 0u};
        uint32_t       h_rng[2]    = {0, 1};
        if (e == hipSuccess) e = hipMemcpyAsync(rng, init_rng, 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cob_ax_range_kernel, dim3(1024), dim3(256), 0, s, nnz, static_cast<const uint32_t*>(d_Ax), rng);
            e = hipMemcpyAsync(h_rng, rng, 8, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail("value range");
        g->x_uniform = h_rng[0] == h_rng[1];
        g->x_bits    = h_rng[0];
    }
    plan->cob = g;
    return SPLAHIP_OK;
}

extern "C" int splahip_csr_plan_cob_info(splahip_plan plan, uint32_t* n_tiles, uint32_t* n_chunks, uint32_t* n_slabs, uint32_t* row_bits) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    const splahip_cob_s* g = plan->cob;
    if (n_tiles) *n_tiles = g ? std::max(g->sched[0].n_tiles, g->sched[1].n_tiles) : 0;
    if (n_chunks) *n_chunks = g ? g->n_chunks : 0;
    if (n_slabs) *n_slabs = g ? std::max(g->sched[0].n_slabs, g->sched[1].n_slabs) : 0;
    if (row_bits) *row_bits = g ? g->row_bits : 0;
    return SPLAHIP_OK;
}
