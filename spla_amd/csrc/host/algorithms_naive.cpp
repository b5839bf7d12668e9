// algorithms_naive.cpp — the plain host-side CHECKERS that the reference's algorithm API ships next to bfs / sssp / pr / tc
// (include/spla/algorithm.hpp:70-84,103-118,139-155,172-175; used by its examples to verify accelerator results).  They work on
// std::vector adjacency lists handed in by the caller and are not a compute path of exec_*: nothing in the operator API ever
// dispatches to them.  Written from the documented semantics (level numbers from 1, FLT_MAX = unreachable, the same power
// iteration and stopping rule as pr(), per-edge sorted-list intersection for tc).
#include "../../../include/spla.hpp"

#include <cmath>
#include <deque>
#include <limits>

namespace spla {

    Status bfs_naive(std::vector<int>& v, std::vector<std::vector<spla::uint>>& A, uint s, const ref_ptr<Descriptor>&) {
        v.assign(v.size(), 0);
        if (s >= v.size()) return Status::InvalidArgument;
        std::deque<uint> todo{s};
        v[s] = 1;
        while (!todo.empty()) {
            const uint at = todo.front();
            todo.pop_front();
            for (uint next : A[at])
                if (v[next] == 0) {
                    v[next] = v[at] + 1;
                    todo.push_back(next);
                }
        }
        return Status::Ok;
    }

    Status sssp_naive(std::vector<float>& v, std::vector<std::vector<uint>>& Ai, std::vector<std::vector<float>>& Ax, uint s, const ref_ptr<Descriptor>&) {
        const float unreachable = std::numeric_limits<float>::max();
        v.assign(v.size(), unreachable);
        if (s >= v.size()) return Status::InvalidArgument;
        std::vector<char> queued(v.size(), 0);
        std::deque<uint>  todo{s};
        v[s]      = 0.0f;
        queued[s] = 1;
        while (!todo.empty()) {// label-correcting relaxation until no distance improves
            const uint at = todo.front();
            todo.pop_front();
            queued[at] = 0;
            for (std::size_t k = 0; k < Ai[at].size(); ++k) {
                const uint  to   = Ai[at][k];
                const float cand = v[at] + Ax[at][k];
                if (v[to] == unreachable || cand < v[to]) {
                    v[to] = cand;
                    if (!queued[to]) {
                        queued[to] = 1;
                        todo.push_back(to);
                    }
                }
            }
        }
        return Status::Ok;
    }

    Status pr_naive(std::vector<float>& p, std::vector<std::vector<uint>>& Ai, std::vector<std::vector<float>>& Ax, float alpha, float eps,
                    const ref_ptr<Descriptor>&) {
        const std::size_t  n        = p.size();
        const float        teleport = (1.0f - alpha) / float(n);
        std::vector<float> last(n, 1.0f / float(n));
        float              step = eps + 0.1f;
        while (step > eps) {
            float norm2 = 0.0f;
            for (std::size_t i = 0; i < n; ++i) {
                float acc = 0.0f;
                for (std::size_t k = 0; k < Ai[i].size(); ++k) acc += Ax[i][k] * last[Ai[i][k]];
                p[i] = acc + teleport;
            }
            for (std::size_t i = 0; i < n; ++i) norm2 += (p[i] - last[i]) * (p[i] - last[i]);
            step = std::sqrt(norm2);
            p.swap(last);
        }
        p.swap(last);// the newest iterate sits in `last` after the loop's final swap
        return Status::Ok;
    }

    Status tc_naive(int& ntrins, std::vector<std::vector<spla::uint>>& Ai, const ref_ptr<Descriptor>&) {
        ntrins = 0;
        for (const auto& mine : Ai)
            for (uint other : mine) {
                const auto& theirs = Ai[other];
                std::size_t a = 0, b = 0;
                while (a < mine.size() && b < theirs.size()) {
                    if (mine[a] == theirs[b]) {
                        ++ntrins;
                        ++a;
                        ++b;
                    } else if (mine[a] < theirs[b]) {
                        ++a;
                    } else {
                        ++b;
                    }
                }
            }
        return Status::Ok;
    }

}// namespace spla
