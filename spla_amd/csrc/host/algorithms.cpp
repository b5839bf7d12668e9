// algorithms.cpp — the four graph algorithms of the reference's algorithm API, written over exec_* only, so every step runs
// on the HIP accelerator and a traversal never leaves the device between two hot-path calls.
// Same operator sequence, semirings, masks and push / pull rule as the reference (src/algorithm.cpp:45-120 bfs, 158-229 sssp,
// 278-335 pr, 380-414 tc); any failing step ends the algorithm with that step's status.
#include "internal.hpp"

#include <cmath>
#include <limits>
#include <utility>

namespace spla {

    namespace {
        struct Direction {
            bool  push, pull, push_pull;
            float front_factor;
            explicit Direction(const ref_ptr<Descriptor>& d) {
                ref_ptr<Descriptor> desc = d ? d : Descriptor::make();// C callers may pass NULL: defaults (push_pull, 10 %)
                push                     = desc->get_push_only();
                pull                     = desc->get_pull_only();
                push_pull                = desc->get_push_pull();
                front_factor             = desc->get_front_factor();
                if (!(push || pull || push_pull)) push = true;
            }
            // push while the frontier holds at most front_factor of the vertices (reference src/algorithm.cpp:93-96)
            bool use_push(int front_size, uint n) const {
                const float density = float(front_size) / float(n);
                return push || (push_pull && density <= front_factor);
            }
        };
#define SPLA_STEP(call)                          \
    do {                                         \
        const Status _st = (call);               \
        if (_st != Status::Ok) return _st;       \
    } while (0)
    }// namespace

    // v[i] = BFS level of vertex i counted from 1 at the source, 0 for unreached vertices
    Status bfs(const ref_ptr<Vector>& v, const ref_ptr<Matrix>& A, uint s, const ref_ptr<Descriptor>& descriptor) {
        if (!v || !A || s >= v->get_n_rows()) return Status::InvalidArgument;
        const uint      n = v->get_n_rows();
        const Direction dir(descriptor);

        ref_ptr<Vector> front      = Vector::make(n, INT);
        ref_ptr<Vector> front_next = Vector::make(n, INT);
        ref_ptr<Scalar> front_size = Scalar::make_int(1);
        ref_ptr<Scalar> level      = Scalar::make_int(1);
        ref_ptr<Scalar> zero       = Scalar::make_int(0);
        ref_ptr<Descriptor> step   = Descriptor::make();
        step->set_early_exit(true);// the first visited in-neighbour settles a vertex
        step->set_struct_only(true);

        SPLA_STEP(front->set_int(s, 1));
        for (int current = 1;; ++current) {
            SPLA_STEP(level->set_int(current));
            SPLA_STEP(exec_v_assign_masked(v, front, level, SECOND_INT, NQZERO_INT));// v[front] = level
            if (dir.use_push(front_size->as_int(), n)) SPLA_STEP(exec_vxm_masked(front_next, v, front, A, BAND_INT, BOR_INT, EQZERO_INT, zero, step));
            else
                SPLA_STEP(exec_mxv_masked(front_next, v, A, front, BAND_INT, BOR_INT, EQZERO_INT, zero, step));
            SPLA_STEP(exec_v_count_mf(front_size, front_next));
            if (front_size->as_int() == 0) break;
            std::swap(front, front_next);
        }
        return Status::Ok;
    }

    // v[i] = length of the shortest path from s, FLT_MAX ("infinity", also v's fill value) for unreachable vertices
    Status sssp(const ref_ptr<Vector>& v, const ref_ptr<Matrix>& A, uint s, const ref_ptr<Descriptor>& descriptor) {
        if (!v || !A || s >= v->get_n_rows()) return Status::InvalidArgument;
        const uint      n   = v->get_n_rows();
        const float     inf = std::numeric_limits<float>::max();
        const Direction dir(descriptor);

        ref_ptr<Vector> no_mask  = Vector::make(n, FLOAT);
        ref_ptr<Vector> relaxed  = Vector::make(n, FLOAT);
        ref_ptr<Vector> changed  = Vector::make(n, FLOAT);
        ref_ptr<Scalar> n_change = Scalar::make_int(0);
        ref_ptr<Scalar> infinity = Scalar::make_float(inf);

        SPLA_STEP(v->set_fill_value(infinity));
        SPLA_STEP(changed->set_fill_value(infinity));
        SPLA_STEP(relaxed->set_fill_value(infinity));
        SPLA_STEP(v->set_float(s, 0.0f));
        SPLA_STEP(changed->set_float(s, 0.0f));

        for (;;) {
            // relax the out-edges of the vertices whose distance changed in the previous round (min-plus)
            if (dir.use_push(n_change->as_int(), n)) SPLA_STEP(exec_vxm_masked(relaxed, no_mask, changed, A, PLUS_FLOAT, MIN_FLOAT, ALWAYS_FLOAT, infinity));
            else
                SPLA_STEP(exec_mxv_masked(relaxed, no_mask, A, changed, PLUS_FLOAT, MIN_FLOAT, ALWAYS_FLOAT, infinity));
            SPLA_STEP(exec_v_eadd_fdb(v, relaxed, changed, MIN_FLOAT));// v = min(v, relaxed); changed = the improved entries
            SPLA_STEP(exec_v_count_mf(n_change, changed));
            if (n_change->as_int() == 0) break;
        }
        return Status::Ok;
    }

    // power iteration p <- A p + (1 - alpha) / N until the Euclidean step length drops to eps; A is expected to be already scaled by
    // alpha / out-degree (reference examples/pr.cpp:81-88)
    Status pr(ref_ptr<Vector>& p, const ref_ptr<Matrix>& A, float alpha, float eps, const ref_ptr<Descriptor>& descriptor) {
        if (!p || !A) return Status::InvalidArgument;
        (void) descriptor;
        const uint n = p->get_n_rows();

        ref_ptr<Vector> no_mask  = Vector::make(n, FLOAT);
        ref_ptr<Vector> previous = Vector::make(n, FLOAT);
        ref_ptr<Vector> product  = Vector::make(n, FLOAT);
        ref_ptr<Vector> teleport = Vector::make(n, FLOAT);
        ref_ptr<Vector> diff2    = Vector::make(n, FLOAT);
        ref_ptr<Scalar> norm2    = Scalar::make(FLOAT);
        ref_ptr<Scalar> zero     = Scalar::make_float(0.0f);

        SPLA_STEP(teleport->fill_with(Scalar::make_float((1.0f - alpha) / float(n))));
        SPLA_STEP(previous->fill_with(Scalar::make_float(1.0f / float(n))));

        float step = eps + 0.1f;
        while (step > eps) {
            SPLA_STEP(exec_mxv_masked(product, no_mask, A, previous, MULT_FLOAT, PLUS_FLOAT, ALWAYS_FLOAT, zero));
            SPLA_STEP(exec_v_eadd(p, product, teleport, PLUS_FLOAT));
            SPLA_STEP(exec_v_eadd(diff2, p, previous, MINUS_POW2_FLOAT));
            SPLA_STEP(exec_v_reduce(norm2, zero, diff2, PLUS_FLOAT));
            step = std::sqrt(norm2->as_float());
            std::swap(p, previous);
        }
        std::swap(p, previous);// the loop's last swap moved the newest iterate into `previous`
        return Status::Ok;
    }

    // triangles of the graph whose strictly-lower-triangular adjacency is A: B = (A x A^T) .* A, ntrins = sum(B)
    Status tc(int& ntrins, const ref_ptr<Matrix>& A, const ref_ptr<Matrix>& B, const ref_ptr<Descriptor>& descriptor) {
        if (!A || !B) return Status::InvalidArgument;
        (void) descriptor;
        ref_ptr<Scalar> zero  = Scalar::make_int(0);
        ref_ptr<Scalar> total = Scalar::make(INT);
        SPLA_STEP(exec_mxmT_masked(B, A, A, A, MULT_INT, PLUS_INT, GTZERO_INT, zero));
        SPLA_STEP(exec_m_reduce(total, zero, B, PLUS_INT));
        ntrins = total->as_int();
        return Status::Ok;
    }

}// namespace spla
