// A user-defined op through the C++ operator API, the way a reference user writes it (reference tests/test_op.cpp:56-59 makes the op,
// tests/test_mxv.cpp:33-89 is the product): exec_mxv_masked with OpBinary::make_float / OpSelect::make_int ops runs ON THE DEVICE —
// the body text is compiled at run time (spla_amd/csrc/kernels/jit.hip) — and must equal the host evaluation of the same std::function.
// Built by oracle/Makefile (target usertests) against include/spla.hpp + libspla_amd.so; run by tests/test_user_ops_gpu.py.
#include <spla.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main() {
    spla::Library::get()->set_accelerator(spla::AcceleratorType::Hip);

    // ---- fp32: custom_plus as op_add (not associative: the device must fold left to right like the reference), built-in MULT ----
    auto custom_plus = spla::OpBinary::make_float("custom_plus", "(float a, float b) { return 0.25f * a + 0.75f * b; }",
                                                  [](float a, float b) { return 0.25f * a + 0.75f * b; });
    auto custom_mult = spla::OpBinary::make_float("custom_mult", "(float a, float b) { return a * b + 1.0f; }", [](float a, float b) { return a * b + 1.0f; });
    auto custom_sel  = spla::OpSelect::make_float("custom_sel", "(float a) { return a > 0.5f; }", [](float a) { return a > 0.5f; });
    const spla::uint M = 257, N = 300;
    auto             A = spla::Matrix::make(M, N, spla::FLOAT);
    auto             v = spla::Vector::make(N, spla::FLOAT);
    auto             mask = spla::Vector::make(M, spla::FLOAT);
    auto             r    = spla::Vector::make(M, spla::FLOAT);
    std::vector<std::vector<std::pair<spla::uint, float>>> rows(M);
    std::vector<float>                                     hv(N), hm(M);
    unsigned                                               seed = 12345;
    auto                                                   rnd  = [&]() { return (seed = seed * 1664525u + 1013904223u) >> 8; };
    for (spla::uint j = 0; j < N; ++j) {
        hv[j] = (rnd() % 64) / 16.0f;
        v->set_float(j, hv[j]);
    }
    for (spla::uint i = 0; i < M; ++i) {
        hm[i] = (rnd() % 4) / 4.0f;
        mask->set_float(i, hm[i]);
        const unsigned deg = i == 7 ? 250 : rnd() % 9;// one long row
        spla::uint     j   = rnd() % 3;
        for (unsigned d = 0; d < deg && j < N; ++d, j += 1 + rnd() % (i == 7 ? 1 : 30)) {
            const float x = (rnd() % 32) / 8.0f;
            rows[i].push_back({j, x});
            A->set_float(i, j, x);
        }
    }
    for (int early = 0; early < 2; ++early) {
        auto desc = spla::Descriptor::make();
        desc->set_early_exit(early != 0);
        const float init = 0.5f;
        spla::Status st = spla::exec_mxv_masked(r, mask, A, v, custom_mult, custom_plus, custom_sel, spla::Scalar::make_float(init), desc);
        CHECK(st == spla::Status::Ok);
        for (spla::uint i = 0; i < M; ++i) {
            float want = init;
            if (hm[i] > 0.5f)
                for (auto& e : rows[i]) {
                    want = 0.25f * want + 0.75f * (e.second * hv[e.first] + 1.0f);
                    if (early && want != init) break;
                }
            float got = -1.0f;
            CHECK(r->get_float(i, got) == spla::Status::Ok);
            if (got != want) {
                std::fprintf(stderr, "row %u early=%d: got %.9g want %.9g\n", i, early, got, want);
                return 1;
            }
        }
    }

    // ---- int32: a user-defined op_add beside the built-in PLUS / EQZERO (saturating add: min(a + b, 1000)) ----
    auto sat_add = spla::OpBinary::make_int("sat_add", "(int a, int b) { return a + b > 1000 ? 1000 : a + b; }", [](int a, int b) { return a + b > 1000 ? 1000 : a + b; });
    auto Ai      = spla::Matrix::make(4, 5, spla::INT);
    auto vi      = spla::Vector::make(5, spla::INT);
    auto mi      = spla::Vector::make(4, spla::INT);
    auto ri      = spla::Vector::make(4, spla::INT);
    const int vals[5] = {300, 0, 300, 0, 700};
    for (int j = 0; j < 5; ++j) vi->set_int(j, vals[j]);
    const int mk[4] = {1, 0, 1, 0};
    for (int i = 0; i < 4; ++i) mi->set_int(i, mk[i]);
    Ai->set_int(1, 0, 2), Ai->set_int(1, 2, 1), Ai->set_int(1, 4, 1), Ai->set_int(3, 4, 1), Ai->set_int(0, 1, 5);
    CHECK(spla::exec_mxv_masked(ri, mi, Ai, vi, spla::MULT_INT, sat_add, spla::EQZERO_INT, spla::Scalar::make_int(0)) == spla::Status::Ok);
    int got[4];
    for (int i = 0; i < 4; ++i) CHECK(ri->get_int(i, got[i]) == spla::Status::Ok);
    CHECK(got[0] == 0 && got[1] == 1000 && got[2] == 0 && got[3] == 700);

    // ---- a body that does not compile: the reference's Status::CompilationError, not a crash ----
    auto broken = spla::OpBinary::make_int("broken", "(int a, int b) { return a +* b; }", [](int a, int b) { return a + b; });
    CHECK(spla::exec_mxv_masked(ri, mi, Ai, vi, spla::MULT_INT, broken, spla::EQZERO_INT, spla::Scalar::make_int(0)) == spla::Status::CompilationError);
    std::printf("USER-OPS OK\n");
    return 0;
}
