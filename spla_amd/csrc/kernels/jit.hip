// jit.hip — user-defined ops on the device: a masked SpMV compiled at run time with hiprtc from the ops' source text.
//
// The reference's user-defined ops carry their body as text — OpBinary::make_float("custom_plus", "(float a, float b) { return 0.25f * a +
// 0.75f * b; }", fn), reference src/op.cpp:294-317, tests/test_op.cpp:56-59 — which its OpenCL backend splices into kernel source and builds at
// run time (src/opencl/cl_program_builder.cpp:65-120); without an accelerator it falls back to the host function (src/core/dispatcher.cpp:55-58).
// This library has no CPU path, so hiprtc is what keeps "arbitrary (op_add, op_mult) semirings" true: the body text is valid HIP device code.
//
// One kernel form serves every combination: a lane per row, the reference's strict left fold from init with its early-exit rule
// (src/cpu/cpu_mxv.hpp:88-103) — exact for any op, associative or not; built-in ops that appear beside a user-defined one are emitted from
// the same expressions ops.hpp evaluates (two's-complement int arithmetic, std::min / std::max comparisons, guarded integer division).
// Compiled kernels are cached by their full source text.  hiprtc is loaded at run time from the directory of the HIP runtime the process
// already uses (PyTorch-ROCm bundles its own), then from the default search path.
#include "common.hpp"
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

    std::string mxv_source(int type, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src) {
        std::string s;
        s += "typedef unsigned int uint;\n";
        s += std::string("typedef ") + type_text(type) + " T;\n";
        s += binary_fn("op_mult", type, op_mult, mult_src);
        s += binary_fn("op_add", type, op_add, add_src);
        s += select_fn(op_select, select_src);
        // reference src/cpu/cpu_mxv.hpp:88-103: every row is written; a selected row folds its products left to right from init and, with
        // early exit, stops after the product that makes the sum differ from init
        s += "extern \"C\" __global__ void __launch_bounds__(256) splahip_mxv_jit(uint n_rows, const uint* __restrict__ Ap, const uint* __restrict__ Aj,\n"
             "        const T* __restrict__ Ax, const T* __restrict__ mask, const T* __restrict__ v, T* __restrict__ r, T init, int early_exit) {\n"
             "    const uint i = blockIdx.x * 256u + threadIdx.x;\n"
             "    if (i >= n_rows) return;\n"
             "    T sum = init;\n"
             "    if (op_select(mask[i])) {\n"
             "        const uint k1 = Ap[i + 1];\n"
             "        for (uint k = Ap[i]; k < k1; ++k) {\n"
             "            sum = op_add(sum, op_mult(Ax[k], v[Aj[k]]));\n"
             "            if (early_exit && sum != init) break;\n"
             "        }\n"
             "    }\n"
             "    r[i] = sum;\n"
             "}\n";
        return s;
    }

    std::mutex                                     g_jit_lock;
    std::unordered_map<std::string, hipFunction_t> g_jit_cache;// full source text -> kernel (modules stay loaded for the life of the process)

    int compile(const std::string& src, hipFunction_t* fn) {
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
        const char*  opts[] = {"--offload-arch=gfx950", "-O3"};
        hiprtcResult cr     = r.CompileProgram(prog, 2, opts);
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
        SPLAHIP_TRY(hipModuleGetFunction(&f, mod, "splahip_mxv_jit"));
        std::lock_guard<std::mutex> guard(g_jit_lock);
        g_jit_cache.emplace(src, f);
        *fn = f;
        return SPLAHIP_OK;
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
    st               = compile(mxv_source(type, op_mult, mult_src, op_add, add_src, op_select, select_src), &fn);
    if (st != SPLAHIP_OK) return st;
    uint32_t init = init_bits;
    int      ee   = early_exit ? 1 : 0;
    void*    args[] = {&n_rows, &d_Ap, &d_Aj, &d_Ax, &d_mask, &d_v, &d_r, &init, &ee};
    SPLAHIP_TRY(hipModuleLaunchKernel(fn, (n_rows + 255u) / 256u, 1, 1, 256, 1, 1, 0, as_stream(stream), args, nullptr));
    return SPLAHIP_OK;
}
}
