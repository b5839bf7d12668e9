// jit.hip — user-defined ops on the device: a masked SpMV compiled at run time with hiprtc from the ops' source text.
//
// The reference's user-defined ops carry their body as text — OpBinary::make_float("custom_plus", "(float a, float b) { return 0.25f * a +
// 0.75f * b; }", fn), reference src/op.cpp:294-317, tests/test_op.cpp:56-59 — which its OpenCL backend splices into kernel source and builds at
// run time (src/opencl/cl_program_builder.cpp:65-120); without an accelerator it falls back to the host function (src/core/dispatcher.cpp:55-58).
// This library has no CPU path, so hiprtc is what keeps "arbitrary (op_add, op_mult) semirings" true: the body text is valid HIP device code.
//
// One kernel form per operator serves every combination, each the reference's own fold order — exact for any op, associative or not:
//   mxv          a lane per row, strict left fold from init with the early-exit rule (src/cpu/cpu_mxv.hpp:88-103)
//   vxm          a lane per output index over the transposed matrix, products in the stored order of v, the first product initialises
//                (src/cpu/cpu_vxm.hpp:92-125)
//   mxm          expand / stable sort / a lane per run folding in expand order from init (src/cpu/cpu_mxm.hpp:85-114)
//   mxmT_masked  a lane per stored mask entry, two-pointer merge of the sorted rows (src/cpu/cpu_mxmT_masked.hpp:87-125)
// The drivers around them are the built-in ops' own (jit_hooks.hpp).  Built-in ops that appear beside a user-defined one are emitted from
// the same expressions ops.hpp evaluates (two's-complement int arithmetic, std::min / std::max comparisons, guarded integer division).
// Device code is compiled with -ffp-contract=off: the op's host std::function — the reference's CPU result — is not fused either.
// Compiled kernels are cached by their full source text.  hiprtc is loaded at run time from the directory of the HIP runtime the process
// already uses (PyTorch-ROCm bundles its own), then from the default search path.
#include "common.hpp"
#include "jit_hooks.hpp"
#include "ops.hpp"

#include <hip/hiprtc.h>

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>

using namespace splahip;

namespace {

    struct Rtc {
        void* lib = nullptr;
        hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
        hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char**)                                          = nullptr;
        hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*)                                                 = nullptr;
        hiprtcResult (*GetProgramLog)(hiprtcProgram, char*)                                                       = nullptr;
        hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*)                                                       = nullptr;
        hiprtcResult (*GetCode)(hiprtcProgram, char*)                                                             = nullptr;
        hiprtcResult (*DestroyProgram)(hiprtcProgram*)                                                            = nullptr;
        bool ok = false;
    };
    Rtc& rtc() {
        static Rtc            r;
        static std::once_flag once;
        std::call_once(once, [] {
            std::string beside;// libhiprtc.so next to the libamdhip64.so this process runs on
            Dl_info     info;
            if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
                beside             = info.dli_fname;
                const size_t slash = beside.rfind('/');
                beside             = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1) + "libhiprtc.so";
            }
            const char* names[] = {beside.c_str(), "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
            for (const char* n : names) {
                if (!n[0]) continue;
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
            }
            if (!r.lib) return;
            auto sym = [&](const char* s) { return dlsym(r.lib, s); };
            r.CreateProgram     = reinterpret_cast<decltype(r.CreateProgram)>(sym("hiprtcCreateProgram"));
            r.CompileProgram    = reinterpret_cast<decltype(r.CompileProgram)>(sym("hiprtcCompileProgram"));
            r.GetProgramLogSize = reinterpret_cast<decltype(r.GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
            r.GetProgramLog     = reinterpret_cast<decltype(r.GetProgramLog)>(sym("hiprtcGetProgramLog"));
            r.GetCodeSize       = reinterpret_cast<decltype(r.GetCodeSize)>(sym("hiprtcGetCodeSize"));
            r.GetCode           = reinterpret_cast<decltype(r.GetCode)>(sym("hiprtcGetCode"));
            r.DestroyProgram    = reinterpret_cast<decltype(r.DestroyProgram)>(sym("hiprtcDestroyProgram"));
            r.ok = r.CreateProgram && r.CompileProgram && r.GetProgramLogSize && r.GetProgramLog && r.GetCodeSize && r.GetCode && r.DestroyProgram;
        });
        return r;
    }

    const char* type_text(int type) { return type == SPLAHIP_INT ? "int" : type == SPLAHIP_UINT ? "uint" : "float"; }

    // the built-in binary ops as HIP expressions over (T a, T b): the same semantics as eval_bin (ops.hpp), reference src/op.cpp:194-241
    std::string builtin_bin(int op, int type) {
        const bool i = type == SPLAHIP_INT, f = type == SPLAHIP_FLOAT;
        switch (op) {
            case SPLAHIP_PLUS: return i ? "(int) ((uint) a + (uint) b)" : "a + b";
            case SPLAHIP_MINUS: return i ? "(int) ((uint) a - (uint) b)" : "a - b";
            case SPLAHIP_MULT: return i ? "(int) ((uint) a * (uint) b)" : "a * b";
            case SPLAHIP_DIV: return f ? "a / b" : i ? "b == 0 ? 0 : ((a == (-2147483647 - 1) && b == -1) ? a : a / b)" : "b == 0 ? 0u : a / b";
            case SPLAHIP_MINUS_POW2: return i ? "(int) (((uint) a - (uint) b) * ((uint) a - (uint) b))" : "(a - b) * (a - b)";
            case SPLAHIP_FIRST: return "a";
            case SPLAHIP_SECOND: return "b";
            case SPLAHIP_BONE: return "(T) 1";
            case SPLAHIP_MIN: return "(b < a) ? b : a";
            case SPLAHIP_MAX: return "(a < b) ? b : a";
            case SPLAHIP_LOR: return "(a || b) ? (T) 1 : (T) 0";
            case SPLAHIP_LAND: return "(a && b) ? (T) 1 : (T) 0";
            case SPLAHIP_BOR: return "a | b";
            case SPLAHIP_BAND: return "a & b";
            case SPLAHIP_BXOR: return "a ^ b";
            default: return "a";
        }
    }
    std::string builtin_sel(int op) {
        switch (op) {
            case SPLAHIP_EQZERO: return "a == (T) 0";
            case SPLAHIP_NQZERO: return "a != (T) 0";
            case SPLAHIP_GTZERO: return "a > (T) 0";
            case SPLAHIP_GEZERO: return "a >= (T) 0";
            case SPLAHIP_LTZERO: return "a < (T) 0";
            case SPLAHIP_LEZERO: return "a <= (T) 0";
            case SPLAHIP_ALWAYS: return "true";
            default: return "false";
        }
    }

    // "__device__ T name(T a, T b) { ... }" from a user body "(T a, T b) { ... }" or from a built-in id
    std::string binary_fn(const char* name, int type, int op, const char* user_src) {
        std::string s = std::string("__device__ __forceinline__ T ") + name;
        if (user_src) return s + user_src + "\n";
        (void) type;
        return s + "(T a, T b) { return " + builtin_bin(op, type) + "; }\n";
    }
    std::string select_fn(int op, const char* user_src) {
        std::string s = "__device__ __forceinline__ bool op_select";
        if (user_src) return s + user_src + "\n";
        return s + "(T a) { return " + builtin_sel(op) + "; }\n";
    }

    std::string prologue(int type, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src, bool with_select = true) {
        std::string s;
        s += "typedef unsigned int uint;\ntypedef unsigned long long u64;\n";
        s += std::string("typedef ") + type_text(type) + " T;\n";
        s += binary_fn("op_mult", type, op_mult, mult_src);
        s += binary_fn("op_add", type, op_add, add_src);
        if (with_select) s += select_fn(op_select, select_src);
        return s;
    }

    // reference src/cpu/cpu_mxmT_masked.hpp:87-125: per stored mask entry (i, j, m): r = init; if select(m): merge the column-sorted rows A[i, :] and
    // B[j, :], r = add(r, mult(a, b)) on equal columns in ascending column order; the entry is kept iff r != init
    std::string mxmT_source(int type, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src) {
        std::string s = prologue(type, op_mult, mult_src, op_add, add_src, op_select, select_src);
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_mxmT_jit(uint n_rows, uint n_entries, const uint* __restrict__ Mp, const uint* __restrict__ Mj,\n"
             "        const T* __restrict__ Mx, const uint* __restrict__ Ap, const uint* __restrict__ Aj, const T* __restrict__ Ax, const uint* __restrict__ Bp,\n"
             "        const uint* __restrict__ Bj, const T* __restrict__ Bx, T* __restrict__ val, uint* __restrict__ keep, T init) {\n"
             "    const uint e = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (e >= n_entries) return;\n"
             "    T r = init;\n"
             "    if (op_select(Mx[e])) {\n"
             "        uint lo = 0, hi = n_rows;\n"
             "        while (hi - lo > 1) { const uint mid = lo + ((hi - lo) >> 1); if (Mp[mid] <= e) lo = mid; else hi = mid; }\n"
             "        const uint j = Mj[e];\n"
             "        uint ka = Ap[lo], ea = Ap[lo + 1], kb = Bp[j], eb = Bp[j + 1];\n"
             "        while (ka < ea && kb < eb) {\n"
             "            const uint ca = Aj[ka], cb = Bj[kb];\n"
             "            if (ca == cb) { r = op_add(r, op_mult(Ax[ka], Bx[kb])); ++ka; ++kb; }\n"
             "            else if (ca < cb) ++ka; else ++kb;\n"
             "        }\n"
             "    }\n"
             "    val[e] = r;\n"
             "    keep[e] = (r != init) ? 1u : 0u;\n"
             "}\n";
        return s;
    }

    // reference src/cpu/cpu_mxm.hpp:85-114: R[i, j] = fold over k in A[i, :] (row order) of add(., mult(a, b)) from init; entries equal to init are dropped.
    // expand: product p of a row block -> (key = row in block << col_bits | column, value = mult(a, b)), in (A entry, B entry) order; the stable sort keeps
    // that order inside a key; run: the lane at a run's head folds it from init
    std::string mxm_source(int type, int op_mult, const char* mult_src, int op_add, const char* add_src) {
        std::string s = prologue(type, op_mult, mult_src, op_add, add_src, 0, nullptr, false);
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_mxm_expand_jit(uint n_products, uint e0, uint n_entries, const uint* __restrict__ offs, uint n_rows,\n"
             "        const uint* __restrict__ Ap, const uint* __restrict__ Aj, const T* __restrict__ Ax, const uint* __restrict__ Bp, const uint* __restrict__ Bj,\n"
             "        const T* __restrict__ Bx, uint r0, uint col_bits, u64* __restrict__ key, T* __restrict__ val) {\n"
             "    const uint p = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (p >= n_products) return;\n"
             "    uint lo = 0, hi = n_entries;\n"
             "    while (hi - lo > 1) { const uint mid = lo + ((hi - lo) >> 1); if (offs[mid] <= p) lo = mid; else hi = mid; }\n"
             "    const uint e = e0 + lo;\n"
             "    uint rl = 0, rh = n_rows;\n"
             "    while (rh - rl > 1) { const uint mid = rl + ((rh - rl) >> 1); if (Ap[mid] <= e) rl = mid; else rh = mid; }\n"
             "    const uint kb = Bp[Aj[e]] + (p - offs[lo]);\n"
             "    key[p] = ((u64) (rl - r0) << col_bits) | (u64) Bj[kb];\n"
             "    val[p] = op_mult(Ax[e], Bx[kb]);\n"
             "}\n"
             "extern \"C\" __global__ void __launch_bounds__(256) splahip_mxm_run_jit(const u64* __restrict__ key, const T* __restrict__ val, uint n, const uint* __restrict__ head,\n"
             "        const uint* __restrict__ run_rank, u64* __restrict__ run_key, T* __restrict__ run_val, uint* __restrict__ run_keep, T init) {\n"
             "    const uint p = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (p >= n || head[p] == 0) return;\n"
             "    T acc = init;\n"
             "    uint q = p;\n"
             "    do { acc = op_add(acc, val[q]); ++q; } while (q < n && head[q] == 0);\n"
             "    const uint rr = run_rank[p];\n"
             "    run_key[rr] = key[p];\n"
             "    run_val[rr] = acc;\n"
             "    run_keep[rr] = (acc != init) ? 1u : 0u;\n"
             "}\n";
        return s;
    }

    // reference src/cpu/cpu_vxm.hpp:92-125: for every stored entry (i, v_i) of v in stored order, for (j, x) in M[i, :] with select(mask[j]): the first product
    // of column j initialises r[j] = mult(v_i, x), later ones fold r[j] = add(r[j], mult(v_i, x)); init is never used.  Pull form: column j of M is row j of the
    // transposed matrix, whose entries ascend in i — the stored order of v when v is index-sorted; otherwise the products are taken by ascending position in v
    std::string vxm_source(int type, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src) {
        std::string s = prologue(type, op_mult, mult_src, op_add, add_src, op_select, select_src);
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_vxm_jit(uint n_out, const uint* __restrict__ Tp, const uint* __restrict__ Tj, const T* __restrict__ Tx,\n"
             "        const T* __restrict__ mask, const T* __restrict__ v, const uint* __restrict__ present, const uint* __restrict__ vpos, int sorted,\n"
             "        T* __restrict__ outv, uint* __restrict__ touched, uint long_from) {\n"
             "    const uint j = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (j >= n_out) return;\n"
             "    if (sorted && Tp[j + 1] - Tp[j] > long_from) return;\n"// (a long column of an index-sorted v: splahip_vxm_long_jit)
             "    T acc = (T) 0;\n"
             "    uint t = 0;\n"
             "    if (op_select(mask[j])) {\n"
             "        const uint k0 = Tp[j], k1 = Tp[j + 1];\n"
             "        if (sorted) {\n"
             "            for (uint k = k0; k < k1; ++k) {\n"
             "                const uint i = Tj[k];\n"
             "                if ((present[i >> 5] >> (i & 31u)) & 1u) { const T p = op_mult(v[i], Tx[k]); acc = t ? op_add(acc, p) : p; t = 1; }\n"
             "            }\n"
             "        } else {\n"
             "            long long last = -1;\n"
             "            for (;;) {\n"
             "                uint best = 0xFFFFFFFFu, bk = 0;\n"
             "                for (uint k = k0; k < k1; ++k) {\n"
             "                    const uint i = Tj[k];\n"
             "                    if ((present[i >> 5] >> (i & 31u)) & 1u) { const uint pos = vpos[i]; if ((long long) pos > last && pos < best) { best = pos; bk = k; } }\n"
             "                }\n"
             "                if (best == 0xFFFFFFFFu) break;\n"
             "                const T p = op_mult(v[Tj[bk]], Tx[bk]);\n"
             "                acc = t ? op_add(acc, p) : p;\n"
             "                t = 1;\n"
             "                last = (long long) best;\n"
             "            }\n"
             "        }\n"
             "    }\n"
             "    outv[j] = acc;\n"
             "    touched[j] = t;\n"
             "}\n";
        return s;
    }

    // The long columns of an index-sorted vxm call with user-defined ops (round 6): one wavefront per listed output index, 64 entries of the transposed matrix's row a
    // step; the running (value, "a product has been seen") pair travels down the lanes (DPP wave_shr:1) — lane t applies op_add to its left neighbour's value and its
    // own product if its index of v is stored, initialises with its product if nothing was seen so far, passes the pair on otherwise — the reference's fold in the
    // stored order of v (src/cpu/cpu_vxm.hpp:101-113) for ANY op.  Until round 6 such a column was one lane's.
    std::string vxm_long_source(int type, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src) {
        std::string s = prologue(type, op_mult, mult_src, op_add, add_src, op_select, select_src);
        s += type == SPLAHIP_FLOAT ? "#define TO_I(x) __float_as_int(x)\n#define FROM_I(x) __int_as_float(x)\n" : "#define TO_I(x) ((int) (x))\n#define FROM_I(x) ((T) (x))\n";
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_vxm_long_jit(const uint* __restrict__ rows, const uint* __restrict__ n_listed, const uint* __restrict__ Tp,\n"
             "        const uint* __restrict__ Tj, const T* __restrict__ Tx, const T* __restrict__ mask, const T* __restrict__ v, const uint* __restrict__ present,\n"
             "        T* __restrict__ outv, uint* __restrict__ touched) {\n"
             "    const uint lane = threadIdx.x & 63u;\n"
             "    const uint n = *n_listed;\n"
             "    for (uint w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n; w += gridDim.x * 4u) {\n"
             "        const uint j = rows[w];\n"
             "        if (!op_select(mask[j])) { if (lane == 0) { outv[j] = (T) 0; touched[j] = 0; } continue; }\n"
             "        const uint rs = Tp[j], re = Tp[j + 1];\n"
             "        T acc = (T) 0;\n"
             "        int has = 0;\n"
             "        for (uint k0 = rs; k0 < re; k0 += 64u) {\n"
             "            const uint k = k0 + lane;\n"
             "            bool ok = false;\n"
             "            T p = (T) 0;\n"
             "            if (k < re) { const uint i = Tj[k]; ok = ((present[i >> 5] >> (i & 31u)) & 1u) != 0u; if (ok) p = op_mult(v[i], Tx[k]); }\n"
             "            if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;\n"
             "            T a = acc;\n"
             "            int h = has;\n"
             "            for (int t = 0; t < 64; ++t) {\n"
             "                const T pa = FROM_I(__builtin_amdgcn_update_dpp(TO_I(acc), TO_I(a), 0x138, 0xF, 0xF, false));\n"
             "                const int ph = __builtin_amdgcn_update_dpp(has, h, 0x138, 0xF, 0xF, false);\n"
             "                if (ok) { a = ph ? op_add(pa, p) : p; h = 1; } else { a = pa; h = ph; }\n"
             "            }\n"
             "            acc = FROM_I(__builtin_amdgcn_readlane(TO_I(a), 63));\n"
             "            has = __builtin_amdgcn_readlane(h, 63);\n"
             "        }\n"
             "        if (lane == 0) { outv[j] = acc; touched[j] = (uint) has; }\n"
             "    }\n"
             "}\n";
        return s;
    }

    std::string mxv_source(int type, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src) {
        std::string s = prologue(type, op_mult, mult_src, op_add, add_src, op_select, select_src);
        // reference src/cpu/cpu_mxv.hpp:88-103: every row is written; a selected row folds its products left to right from init and, with
        // early exit, stops after the product that makes the sum differ from init.  Rows of more than `long_from` entries are left to splahip_mxv_long_jit below
        // (0xFFFFFFFF: none — the early-exit calls, whose rows mostly stop after a few entries).
        s += type == SPLAHIP_FLOAT ? "#define TO_I(x) __float_as_int(x)\n#define FROM_I(x) __int_as_float(x)\n" : "#define TO_I(x) ((int) (x))\n#define FROM_I(x) ((T) (x))\n";
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_mxv_jit(uint n_rows, const uint* __restrict__ Ap, const uint* __restrict__ Aj,\n"
             "        const T* __restrict__ Ax, const T* __restrict__ mask, const T* __restrict__ v, T* __restrict__ r, T init, int early_exit, uint long_from) {\n"
             "    const uint i = blockIdx.x * 256u + threadIdx.x;\n"
             "    const uint lane = threadIdx.x & 63u;\n"
             "    const bool in = i < n_rows;\n"
             "    const uint k0 = in ? Ap[i] : 0u, k1 = in ? Ap[i + 1] : 0u;\n"
             "    const bool sel = in && op_select(mask[i]);\n"
             "    T sum = init;\n"
             "    if (!early_exit) {\n"
             "        if (!in || k1 - k0 > long_from) return;\n"
             "        if (sel) for (uint k = k0; k < k1; ++k) sum = op_add(sum, op_mult(Ax[k], v[Aj[k]]));\n"
             "        r[i] = sum;\n"
             "        return;\n"
             "    }\n"
             // early exit: a lane walks the first 64 entries of its row; what is left of a longer row is searched by the wave, 64 entries a step (mxv_kernels.hpp:
             // mxv_rowseq_kernel<EARLY> — the same rule, the same handling of a sum whose bits change without it moving off init)
             "    uint k = sel ? k0 : 0u;\n"
             "    const uint kend = sel ? k1 : 0u, ke = kend < k + 64u ? kend : k + 64u;\n"
             "    bool done = false;\n"
             "    for (; k < ke; ++k) { sum = op_add(sum, op_mult(Ax[k], v[Aj[k]])); if (sum != init) { done = true; break; } }\n"
             "    u64 todo = __builtin_amdgcn_ballot_w64(sel && !done && ke < kend);\n"
             "    while (todo != 0ull) {\n"
             "        const int src = __builtin_ctzll(todo);\n"
             "        const uint a = (uint) __builtin_amdgcn_readlane((int) ke, src), b = (uint) __builtin_amdgcn_readlane((int) kend, src);\n"
             "        T S = FROM_I(__builtin_amdgcn_readlane(TO_I(sum), src));\n"
             "        bool stopped = false;\n"
             "        for (uint q = a; q < b && !stopped; q += 64u) {\n"
             "            const uint kk = q + lane;\n"
             "            const bool valid = kk < b;\n"
             "            T p = init;\n"
             "            if (valid) p = op_mult(Ax[kk], v[Aj[kk]]);\n"
             "            const T c = valid ? op_add(S, p) : S;\n"
             "            const bool off = valid && c != init;\n"
             "            const bool flip = valid && !off && TO_I(c) != TO_I(S);\n"
             "            const u64 m = __builtin_amdgcn_ballot_w64(off), f = __builtin_amdgcn_ballot_w64(flip);\n"
             "            const u64 before = m != 0ull ? ((m & (0ull - m)) - 1ull) : ~0ull;\n"
             "            if ((f & before) != 0ull) {\n"
             "                T x = S;\n"
             "                for (int t = 0; t < 64; ++t) {\n"
             "                    const T prev = FROM_I(__builtin_amdgcn_update_dpp(TO_I(S), TO_I(x), 0x138, 0xF, 0xF, false));\n"
             "                    x = valid ? op_add(prev, p) : prev;\n"
             "                }\n"
             "                const u64 m2 = __builtin_amdgcn_ballot_w64(valid && x != init);\n"
             "                if (m2 != 0ull) { S = FROM_I(__builtin_amdgcn_readlane(TO_I(x), __builtin_ctzll(m2))); stopped = true; }\n"
             "                else S = FROM_I(__builtin_amdgcn_readlane(TO_I(x), 63));\n"
             "            } else if (m != 0ull) { S = FROM_I(__builtin_amdgcn_readlane(TO_I(c), __builtin_ctzll(m))); stopped = true; }\n"
             "        }\n"
             "        if ((int) lane == src) sum = S;\n"
             "        todo &= todo - 1ull;\n"
             "    }\n"
             "    if (in) r[i] = sum;\n"
             "}\n";
        return s;
    }

    // The long rows of a call with user-defined ops (round 6): one wavefront per listed row, 64 entries a step (coalesced Aj / Ax, gathers of v, the next step's loads
    // in flight); the step's products are added ONE AFTER THE OTHER in entry order — the running value travels down the lanes (DPP wave_shr:1: in round t lane t takes
    // lane t - 1's value and applies op_add to it and its own product; lane t's value is final after round t) — so the result is the reference's sequential fold for
    // ANY op_add, associative or not (src/cpu/cpu_mxv.hpp:94-99).  Until round 6 such a row was one LANE's: a hub row of 10^5 entries, 10^5 dependent gathers.
    std::string mxv_long_source(int type, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src) {
        std::string s = prologue(type, op_mult, mult_src, op_add, add_src, op_select, select_src);
        s += type == SPLAHIP_FLOAT ? "#define TO_I(x) __float_as_int(x)\n#define FROM_I(x) __int_as_float(x)\n" : "#define TO_I(x) ((int) (x))\n#define FROM_I(x) ((T) (x))\n";
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_mxv_long_jit(const uint* __restrict__ rows, const uint* __restrict__ n_listed, const uint* __restrict__ Ap,\n"
             "        const uint* __restrict__ Aj, const T* __restrict__ Ax, const T* __restrict__ mask, const T* __restrict__ v, T* __restrict__ r, T init) {\n"
             "    const uint lane = threadIdx.x & 63u;\n"
             "    const uint n = *n_listed;\n"
             "    for (uint w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n; w += gridDim.x * 4u) {\n"
             "        const uint row = rows[w];\n"
             "        if (!op_select(mask[row])) { if (lane == 0) r[row] = init; continue; }\n"
             "        const uint rs = Ap[row], re = Ap[row + 1];\n"
             "        T acc = init;\n"
             "        uint k = rs + lane;\n"
             "        T p = init;\n"
             "        if (k < re) p = op_mult(Ax[k], v[Aj[k]]);\n"
             "        for (uint k0 = rs; k0 < re; k0 += 64u) {\n"
             "            const T pc = p;\n"
             "            k = k0 + 64u + lane;\n"
             "            if (k < re) p = op_mult(Ax[k], v[Aj[k]]);\n"
             "            const uint cnt = re - k0 < 64u ? re - k0 : 64u;\n"
             "            if (cnt == 64u) {\n"
             "                T a = acc;\n"
             "#pragma unroll 16\n"
             "                for (int t = 0; t < 64; ++t) {\n"
             "                    const T prev = FROM_I(__builtin_amdgcn_update_dpp(TO_I(acc), TO_I(a), 0x138, 0xF, 0xF, false));\n"
             "                    a = op_add(prev, pc);\n"
             "                }\n"
             "                acc = FROM_I(__builtin_amdgcn_readlane(TO_I(a), 63));\n"
             "            } else {\n"
             "                for (uint l = 0; l < cnt; ++l) acc = op_add(acc, FROM_I(__builtin_amdgcn_readlane(TO_I(pc), l)));\n"
             "            }\n"
             "        }\n"
             "        if (lane == 0) r[row] = acc;\n"
             "    }\n"
             "}\n";
        return s;
    }

    // ---- the vector operations around the hot path with user-defined ops: element-wise kernels, one lane per element / stored entry ----------
    // reference src/cpu/cpu_v_assign.hpp:68-132, cpu_v_eadd_fdb.hpp:71-139, cpu_v_eadd.hpp:70-157, cpu_v_reduce.hpp:69-117, cpu_v_map.hpp:57-68
    std::string vector_source(int type, int op, const char* op_src, int op_select, const char* select_src) {
        std::string s = "typedef unsigned int uint;\n";
        s += std::string("typedef ") + type_text(type) + " T;\n";
        s += binary_fn("op_user", type, op, op_src);
        s += select_fn(op_select, select_src);
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_assign_dense_jit(uint n, T* __restrict__ r, const T* __restrict__ mask, T value) {\n"
             "    const uint i = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (i < n && op_select(mask[i])) r[i] = op_user(r[i], value);\n"
             "}\n"
             "extern \"C\" __global__ void __launch_bounds__(256) splahip_assign_sparse_jit(uint nvals, T* __restrict__ r, const uint* __restrict__ mi, const T* __restrict__ mx, T value) {\n"
             "    const uint k = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (k < nvals && op_select(mx[k])) { const uint i = mi[k]; r[i] = op_user(r[i], value); }\n"
             "}\n"
             "extern \"C\" __global__ void __launch_bounds__(256) splahip_eadd_fdb_dense_jit(uint n, T* __restrict__ r, const T* __restrict__ v, T* __restrict__ fdb, T fdb_fill) {\n"
             "    const uint i = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (i >= n) return;\n"
             "    const T prev = r[i];\n"
             "    const T next = op_user(prev, v[i]);\n"
             "    r[i]   = next;\n"
             "    fdb[i] = prev != next ? next : fdb_fill;\n"
             "}\n"
             "extern \"C\" __global__ void __launch_bounds__(256) splahip_eadd_fdb_apply_jit(uint nvals, T* __restrict__ r, const uint* __restrict__ vi, const T* __restrict__ vx,\n"
             "        uint* __restrict__ changed, T* __restrict__ newval) {\n"
             "    const uint k = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (k >= nvals) return;\n"
             "    const uint i    = vi[k];\n"
             "    const T    prev = r[i];\n"
             "    const T    next = op_user(prev, vx[k]);\n"
             "    r[i]       = next;\n"
             "    changed[k] = prev != next ? 1u : 0u;\n"
             "    newval[k]  = next;\n"
             "}\n"
             "extern \"C\" __global__ void __launch_bounds__(256) splahip_eadd_dense_jit(uint n, T* __restrict__ r, const T* __restrict__ u, const T* __restrict__ v) {\n"
             "    const uint i = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (i < n) r[i] = op_user(u[i], v[i]);\n"
             "}\n"
             // the reference's strict left fold from s (a user op need not be associative): 64 lanes fetch 1024 elements, lane 0 folds them in order out of LDS
             "extern \"C\" __global__ void __launch_bounds__(64) splahip_reduce_jit(uint n, const T* __restrict__ x, T s, T* __restrict__ out) {\n"
             "    __shared__ T s_buf[1024];\n"
             "    T sum = s;\n"
             "    for (uint base = 0; base < n; base += 1024u) {\n"
             "        const uint len = n - base < 1024u ? n - base : 1024u;\n"
             "        for (uint k = threadIdx.x; k < len; k += 64u) s_buf[k] = x[base + k];\n"
             "        __syncthreads();\n"
             "        if (threadIdx.x == 0) for (uint k = 0; k < len; ++k) sum = op_user(sum, s_buf[k]);\n"
             "        __syncthreads();\n"
             "    }\n"
             "    if (threadIdx.x == 0) out[0] = sum;\n"
             "}\n";
        return s;
    }
    std::string map_source(int type, const char* op_src) {
        std::string s = "typedef unsigned int uint;\n";
        s += std::string("typedef ") + type_text(type) + " T;\n";
        s += std::string("__device__ __forceinline__ T op_apply") + op_src + "\n";
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_map_jit(uint n, const T* __restrict__ x, T* __restrict__ out) {\n"
             "    const uint i = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (i < n) out[i] = op_apply(x[i]);\n"
             "}\n";
        return s;
    }

    std::mutex                                     g_jit_lock;
    std::unordered_map<std::string, hipFunction_t> g_jit_cache;// full source text -> kernel (modules stay loaded for the life of the process)

    int compile(const std::string& source, const char* kernel_name, hipFunction_t* fn) {
        const std::string src = std::string("// ") + kernel_name + "\n" + source;// (one cache entry per kernel of a source)
        {
            std::lock_guard<std::mutex> guard(g_jit_lock);
            auto                        it = g_jit_cache.find(src);
            if (it != g_jit_cache.end()) {
                *fn = it->second;
                return SPLAHIP_OK;
            }
        }
        Rtc& r = rtc();
        if (!r.ok) return fail(SPLAHIP_NOT_IMPLEMENTED, "hiprtc is not loadable in this process: user-defined ops cannot be compiled for the device");
        hiprtcProgram prog = nullptr;
        if (r.CreateProgram(&prog, src.c_str(), "splahip_user_ops.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return fail(SPLAHIP_ERROR, "hiprtcCreateProgram failed");
        // the architecture of the device this process runs on (gcnArchName, e.g. "gfx950:sramecc+:xnack-"); no fused multiply-add across the user's
        // expression: the op's host function, which stands in for the reference's CPU result, is not contracted either
        std::string     arch = "--offload-arch=gfx950";
        int             dev  = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
            std::string name = prop.gcnArchName;
            arch             = "--offload-arch=" + name.substr(0, name.find(':'));
        }
        const char*  opts[] = {arch.c_str(), "-O3", "-ffp-contract=off"};
        hiprtcResult cr     = r.CompileProgram(prog, 3, opts);
        if (cr != HIPRTC_SUCCESS) {
            size_t      n = 0;
            std::string log;
            if (r.GetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
                log.resize(n);
                (void) r.GetProgramLog(prog, &log[0]);
            }
            (void) r.DestroyProgram(&prog);
            if (log.size() > 700) log.resize(700);
            return fail(8 /* spla Status::CompilationError, reference include/spla/config.hpp:80 */, "user-defined op does not compile for the device: %s", log.c_str());
        }
        size_t bytes = 0;
        (void) r.GetCodeSize(prog, &bytes);
        std::string code(bytes, '\0');
        (void) r.GetCode(prog, &code[0]);
        (void) r.DestroyProgram(&prog);
        hipModule_t mod = nullptr;
        SPLAHIP_TRY(hipModuleLoadData(&mod, code.data()));
        hipFunction_t f = nullptr;
        SPLAHIP_TRY(hipModuleGetFunction(&f, mod, kernel_name));
        std::lock_guard<std::mutex> guard(g_jit_lock);
        g_jit_cache.emplace(src, f);
        *fn = f;
        return SPLAHIP_OK;
    }

}// namespace

namespace {
    constexpr uint32_t kJitLongFrom = 508;// rows longer than this go to splahip_mxv_long_jit (the wave-tile plan's long-row threshold)
    // list[1 + i], i < list[0]: the rows of more than long_from entries, in no particular order
    __global__ __launch_bounds__(256) void jit_long_rows_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, uint32_t long_from, uint32_t* __restrict__ list) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i < n_rows && Ap[i + 1] - Ap[i] > long_from) list[1 + atomicAdd(list, 1u)] = i;
    }
}// namespace

extern "C" {

int splahip_jit_available(void) { return rtc().ok ? 1 : 0; }

// *_src: NULL = the built-in op with that id; else the op's body text "(T a, T b) { ... }" / "(T a) { ... }" with T spelled int / uint / float
int splahip_mxv_masked_jit(int type, uint32_t n_rows, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask, const void* d_v, void* d_r,
                           int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src, uint32_t init_bits,
                           int early_exit, void* stream) {
    SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
    SPLAHIP_REQUIRE(mult_src || binop_valid_for(op_mult, type), "op_multiply not defined for this type");
    SPLAHIP_REQUIRE(add_src || binop_valid_for(op_add, type), "op_add not defined for this type");
    SPLAHIP_REQUIRE(select_src || (op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT), "unknown op_select");
    if (n_rows == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_Ap && d_mask && d_r, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipFunction_t fn = nullptr;
    st               = compile(mxv_source(type, op_mult, mult_src, op_add, add_src, op_select, select_src), "splahip_mxv_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    uint32_t init = init_bits;
    int      ee   = early_exit ? 1 : 0;
    uint32_t long_from = early_exit ? 0xFFFFFFFFu : kJitLongFrom;
    void*    args[] = {&n_rows, &d_Ap, &d_Aj, &d_Ax, &d_mask, &d_v, &d_r, &init, &ee, &long_from};
    SPLAHIP_TRY(hipModuleLaunchKernel(fn, (n_rows + 255u) / 256u, 1, 1, 256, 1, 1, 0, as_stream(stream), args, nullptr));
    if (!early_exit) {// the long rows: listed on the device, a wavefront each (mxv_long_source)
        hipFunction_t fl = nullptr;
        st               = compile(mxv_long_source(type, op_mult, mult_src, op_add, add_src, op_select, select_src), "splahip_mxv_long_jit", &fl);
        if (st != SPLAHIP_OK) return st;
        hipStream_t s = as_stream(stream);
        DevMem      list;
        SPLAHIP_TRY(list.alloc(s, sizeof(uint32_t) * ((size_t) n_rows + 1)));
        SPLAHIP_TRY(hipMemsetAsync(list.p, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(jit_long_rows_kernel, dim3((n_rows + 255u) / 256u), dim3(256), 0, s, n_rows, d_Ap, kJitLongFrom, list.as<uint32_t>());
        SPLAHIP_TRY(hipGetLastError());
        const uint32_t* rows  = list.as<uint32_t>() + 1;
        const uint32_t* count = list.as<uint32_t>();
        void*           la[]  = {&rows, &count, &d_Ap, &d_Aj, &d_Ax, &d_mask, &d_v, &d_r, &init};
        SPLAHIP_TRY(hipModuleLaunchKernel(fl, 2048, 1, 1, 256, 1, 1, 0, s, la, nullptr));
    }
    return SPLAHIP_OK;
}

// exec_vxm_masked with user-defined ops: the pull form over the transposed matrix (plan_T = the plan of M^T, n_in = rows of M = length of v, n_out = columns
// of M); sorted: the indices of v ascend (splahip_vxm_probe).  Output: index-sorted COO, one entry per selected column that holds a product.
int splahip_vxm_masked_jit(splahip_plan plan_T, int type, uint32_t n_in, uint32_t n_out, const uint32_t* d_Tp, const uint32_t* d_Tj, const void* d_Tx, const void* d_mask,
                           uint32_t nvals, const uint32_t* d_vi, const void* d_vx, int sorted, uint32_t* d_ri, void* d_rx, uint32_t* h_count, int op_mult, const char* mult_src,
                           int op_add, const char* add_src, int op_select, const char* select_src, void* stream) {
    SPLAHIP_REQUIRE(plan_T != nullptr && h_count != nullptr, "null argument");
    SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
    SPLAHIP_REQUIRE(mult_src || binop_valid_for(op_mult, type), "op_multiply not defined for this type");
    SPLAHIP_REQUIRE(add_src || binop_valid_for(op_add, type), "op_add not defined for this type");
    SPLAHIP_REQUIRE(select_src || (op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT), "unknown op_select");
    *h_count = 0;
    if (n_out == 0 || nvals == 0) return SPLAHIP_OK;
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipFunction_t fn = nullptr;
    st               = compile(vxm_source(type, op_mult, mult_src, op_add, add_src, op_select, select_src), "splahip_vxm_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    hipFunction_t fl = nullptr;// the long columns of an index-sorted v: a wavefront each (vxm_long_source)
    if (sorted) {
        st = compile(vxm_long_source(type, op_mult, mult_src, op_add, add_src, op_select, select_src), "splahip_vxm_long_jit", &fl);
        if (st != SPLAHIP_OK) return st;
    }
    return vxm_pull_jit(plan_T, n_out, n_in, d_Tp, d_Tj, d_Tx, d_mask, nvals, d_vi, d_vx, sorted, d_ri, d_rx, h_count, fn, fl, as_stream(stream));
}

// exec_mxmT_masked with user-defined ops (arguments as splahip_mxmT_masked)
int splahip_mxmT_masked_jit(int type, uint32_t n_rows, uint32_t mask_nnz, const uint32_t* d_Mp, const uint32_t* d_Mj, const void* d_Mx, const uint32_t* d_Ap,
                            const uint32_t* d_Aj, const void* d_Ax, const uint32_t* d_Bp, const uint32_t* d_Bj, const void* d_Bx, uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx,
                            uint32_t* h_nnz, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src, uint32_t init_bits,
                            void* stream) {
    SPLAHIP_REQUIRE(d_Rp && d_Rj && d_Rx && h_nnz, "null output argument");
    SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
    SPLAHIP_REQUIRE(mult_src || binop_valid_for(op_mult, type), "op_multiply not defined for this type");
    SPLAHIP_REQUIRE(add_src || binop_valid_for(op_add, type), "op_add not defined for this type");
    SPLAHIP_REQUIRE(select_src || (op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT), "unknown op_select");
    SPLAHIP_REQUIRE(d_Mp && d_Ap && d_Bp, "null row pointer array");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipFunction_t fn = nullptr;
    st               = compile(mxmT_source(type, op_mult, mult_src, op_add, add_src, op_select, select_src), "splahip_mxmT_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    return mxmT_jit(n_rows, mask_nnz, d_Mp, d_Mj, d_Mx, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, init_bits, fn, as_stream(stream));
}

// exec_mxm with user-defined ops (arguments as splahip_mxm)
int splahip_mxm_jit(int type, uint32_t a_rows, uint32_t b_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const uint32_t* d_Bp, const uint32_t* d_Bj,
                    const void* d_Bx, uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx, uint32_t* h_nnz, int op_mult, const char* mult_src, int op_add, const char* add_src,
                    uint32_t init_bits, void* stream) {
    SPLAHIP_REQUIRE(d_Rp && d_Rj && d_Rx && h_nnz, "null output argument");
    SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
    SPLAHIP_REQUIRE(mult_src || binop_valid_for(op_mult, type), "op_multiply not defined for this type");
    SPLAHIP_REQUIRE(add_src || binop_valid_for(op_add, type), "op_add not defined for this type");
    SPLAHIP_REQUIRE(d_Ap && d_Bp, "null row pointer array");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    const std::string src = mxm_source(type, op_mult, mult_src, op_add, add_src);
    hipFunction_t     expand = nullptr, run = nullptr;
    st = compile(src, "splahip_mxm_expand_jit", &expand);
    if (st == SPLAHIP_OK) st = compile(src, "splahip_mxm_run_jit", &run);
    if (st != SPLAHIP_OK) return st;
    return mxm_jit(a_rows, b_cols, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, init_bits, expand, run, as_stream(stream));
}

// ---- the vector operations with user-defined ops (op_src / select_src: the op's body text; NULL = the built-in op with that id) -----------------------------
namespace {
    int vector_kernel(int type, int op, const char* op_src, int op_select, const char* select_src, const char* name, hipFunction_t* fn) {
        SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
        SPLAHIP_REQUIRE(op_src || binop_valid_for(op, type), "op not defined for this type");
        SPLAHIP_REQUIRE(select_src || (op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT), "unknown op_select");
        int st = ensure_device();
        if (st != SPLAHIP_OK) return st;
        return compile(vector_source(type, op, op_src, op_select, select_src), name, fn);
    }
}// namespace

int splahip_v_assign_masked_jit(int type, int sparse, uint32_t n, void* d_r, const uint32_t* d_mi, const void* d_mask, uint32_t value_bits, int op_assign, const char* assign_src,
                                int op_select, const char* select_src, void* stream) {
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_mask && (!sparse || d_mi), "null device pointer");
    hipFunction_t fn = nullptr;
    int           st = vector_kernel(type, op_assign, assign_src, op_select, select_src, sparse ? "splahip_assign_sparse_jit" : "splahip_assign_dense_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    uint32_t value = value_bits;
    if (sparse) {
        void* args[] = {&n, &d_r, &d_mi, &d_mask, &value};
        SPLAHIP_TRY(hipModuleLaunchKernel(fn, (n + 255u) / 256u, 1, 1, 256, 1, 1, 0, as_stream(stream), args, nullptr));
    } else {
        void* args[] = {&n, &d_r, &d_mask, &value};
        SPLAHIP_TRY(hipModuleLaunchKernel(fn, (n + 255u) / 256u, 1, 1, 256, 1, 1, 0, as_stream(stream), args, nullptr));
    }
    return SPLAHIP_OK;
}

int splahip_v_eadd_fdb_dense_jit(int type, uint32_t n, void* d_r, const void* d_v, void* d_fdb, uint32_t fdb_fill_bits, const char* op_src, void* stream) {
    SPLAHIP_REQUIRE(op_src != nullptr, "op source is null");
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_v && d_fdb, "null device pointer");
    hipFunction_t fn = nullptr;
    int           st = vector_kernel(type, 0, op_src, SPLAHIP_ALWAYS, nullptr, "splahip_eadd_fdb_dense_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    uint32_t fill   = fdb_fill_bits;
    void*    args[] = {&n, &d_r, &d_v, &d_fdb, &fill};
    SPLAHIP_TRY(hipModuleLaunchKernel(fn, (n + 255u) / 256u, 1, 1, 256, 1, 1, 0, as_stream(stream), args, nullptr));
    return SPLAHIP_OK;
}

int splahip_v_eadd_fdb_sparse_jit(int type, uint32_t nvals, void* d_r, const uint32_t* d_vi, const void* d_vx, uint32_t* d_fi, void* d_fx, uint32_t* h_count, const char* op_src,
                                  void* stream) {
    SPLAHIP_REQUIRE(op_src != nullptr && h_count != nullptr, "null argument");
    *h_count = 0;
    if (nvals == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_vi && d_vx && d_fi && d_fx, "null device pointer");
    hipFunction_t fn = nullptr;
    int           st = vector_kernel(type, 0, op_src, SPLAHIP_ALWAYS, nullptr, "splahip_eadd_fdb_apply_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    return eadd_fdb_sparse_jit(nvals, d_r, d_vi, d_vx, d_fi, d_fx, h_count, fn, as_stream(stream));
}

int splahip_v_eadd_dense_jit(int type, uint32_t n, void* d_r, const void* d_u, const void* d_v, const char* op_src, void* stream) {
    SPLAHIP_REQUIRE(op_src != nullptr, "op source is null");
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_u && d_v, "null device pointer");
    hipFunction_t fn = nullptr;
    int           st = vector_kernel(type, 0, op_src, SPLAHIP_ALWAYS, nullptr, "splahip_eadd_dense_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    void* args[] = {&n, &d_r, &d_u, &d_v};
    SPLAHIP_TRY(hipModuleLaunchKernel(fn, (n + 255u) / 256u, 1, 1, 256, 1, 1, 0, as_stream(stream), args, nullptr));
    return SPLAHIP_OK;
}

int splahip_v_eadd_sparse_jit(int type, uint32_t nu, const uint32_t* d_ui, const void* d_ux, uint32_t u_fill_bits, uint32_t nv, const uint32_t* d_vi, const void* d_vx,
                              uint32_t v_fill_bits, uint32_t* d_ri, void* d_rx, uint32_t* h_count, const char* op_src, void* stream) {
    SPLAHIP_REQUIRE(op_src != nullptr && h_count != nullptr, "null argument");
    *h_count = 0;
    if (nu + nv == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_ri && d_rx && (nu == 0 || (d_ui && d_ux)) && (nv == 0 || (d_vi && d_vx)), "null device pointer");
    hipFunction_t fn = nullptr;
    int           st = vector_kernel(type, 0, op_src, SPLAHIP_ALWAYS, nullptr, "splahip_eadd_dense_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    return eadd_sparse_jit(type, nu, d_ui, d_ux, u_fill_bits, nv, d_vi, d_vx, v_fill_bits, d_ri, d_rx, h_count, fn, as_stream(stream));
}

int splahip_reduce_jit(int type, uint32_t n, const void* d_x, uint32_t s_bits, uint32_t* h_result_bits, const char* op_src, void* stream) {
    SPLAHIP_REQUIRE(op_src != nullptr && h_result_bits != nullptr, "null argument");
    *h_result_bits = s_bits;
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_x != nullptr, "null device pointer");
    hipFunction_t fn = nullptr;
    int           st = vector_kernel(type, 0, op_src, SPLAHIP_ALWAYS, nullptr, "splahip_reduce_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    DevMem      out;
    SPLAHIP_TRY(out.alloc(s, 4));
    uint32_t sb     = s_bits;
    void*    po     = out.p;
    void*    args[] = {&n, &d_x, &sb, &po};
    SPLAHIP_TRY(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr));
    SPLAHIP_TRY(fetch_words(h_result_bits, out.p, 1, s));
    return SPLAHIP_OK;
}

int splahip_map_jit(int type, uint32_t n, const void* d_x, void* d_out, const char* op_src, void* stream) {
    SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
    SPLAHIP_REQUIRE(op_src != nullptr, "op source is null");
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_x && d_out, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipFunction_t fn = nullptr;
    st               = compile(map_source(type, op_src), "splahip_map_jit", &fn);
    if (st != SPLAHIP_OK) return st;
    void* args[] = {&n, &d_x, &d_out};
    SPLAHIP_TRY(hipModuleLaunchKernel(fn, (n + 255u) / 256u, 1, 1, 256, 1, 1, 0, as_stream(stream), args, nullptr));
    return SPLAHIP_OK;
}
}
