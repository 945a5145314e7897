// CPU check of spx::for_each_piece / run_of_position (spx_layout.h): the cut rule the flatten step's kernels run on the
// device, held against its specification on random run lists.  Built and run by tests/test_piece_cuts.py (g++).
//   every piece is 1 .. 65535 positions, the pieces tile the run in order;
//   a piece's LF image covers at most `span` runs (span > 0);
//   nothing is cut that did not have to be: a piece that ends before the run does ends because the next position would
//   have been the 65536th or would have begun the (span + 1)-th run.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../spumoni_amd/csrc/spx_layout.h"

using namespace spx;

struct Piece {
    uint64_t off, len;
};

int main() {
    std::mt19937_64 rng(12345);
    uint64_t checked = 0, cut_runs = 0;
    for (int trial = 0; trial < 400; ++trial) {
        const uint64_t r = 2 + rng() % 3000;
        const int shape = trial % 4;
        std::vector<uint64_t> lens(r), S(r + 1);
        for (uint64_t k = 0; k < r; ++k) {
            uint64_t l = 1 + rng() % 5;
            if (shape == 1 && rng() % 20 == 0) l = 60000 + rng() % 200000;       // runs around and past 2^16
            if (shape == 2 && rng() % 50 == 0) l = 1 + rng() % 4000;             // long among short
            if (shape == 3) l = 1 + (uint64_t)(1.0 / (1e-6 + (rng() % 1000000) / 1e6));  // heavy tail
            lens[k] = l;
        }
        S[0] = 0;
        for (uint64_t k = 0; k < r; ++k) S[k + 1] = S[k] + lens[k];
        const uint64_t n = S[r];
        for (uint32_t span : {0u, 1u, 2u, 5u, 16u, 64u}) {
            for (int rep = 0; rep < 40; ++rep) {
                // an image: any interval [lf, lf + len) of [0, n) -- the LF image of a run is one
                const uint64_t len = (rep % 3 == 0) ? lens[rng() % r] : 1 + rng() % (rep % 3 == 1 ? 300 : 300000);
                if (len > n) continue;
                const uint64_t lf = rng() % (n - len + 1);
                const uint64_t a = run_of_position(S.data(), r, lf), b = run_of_position(S.data(), r, lf + len - 1);
                if (!(S[a] <= lf && lf < S[a + 1]) || !(S[b] <= lf + len - 1 && lf + len - 1 < S[b + 1])) {
                    std::printf("run_of_position wrong\n");
                    return 1;
                }
                std::vector<Piece> ps;
                const uint32_t np = for_each_piece(len, lf, span ? S.data() : nullptr, a, b - a, span,
                                                   [&](uint64_t o, uint64_t l) { ps.push_back({o, l}); });
                if (np != ps.size() || ps.empty()) return 2;
                uint64_t pos = 0;
                for (size_t j = 0; j < ps.size(); ++j) {
                    const Piece& p = ps[j];
                    if (p.off != pos || p.len == 0 || p.len > PIECE_MAX) {
                        std::printf("bad piece %zu: off %llu len %llu\n", j, (unsigned long long)p.off, (unsigned long long)p.len);
                        return 3;
                    }
                    const uint64_t first = run_of_position(S.data(), r, lf + p.off), last = run_of_position(S.data(), r, lf + p.off + p.len - 1);
                    if (span && last - first + 1 > span) {
                        std::printf("piece covers %llu runs, span %u\n", (unsigned long long)(last - first + 1), span);
                        return 4;
                    }
                    pos += p.len;
                    if (pos < len) {  // cut here: it had to be
                        const bool full = p.len == PIECE_MAX;
                        const bool at_span = span && run_of_position(S.data(), r, lf + pos) == first + span && S[first + span] == lf + pos;
                        if (!full && !at_span) {
                            std::printf("needless cut at %llu of %llu (span %u)\n", (unsigned long long)pos, (unsigned long long)len, span);
                            return 5;
                        }
                    }
                }
                if (pos != len) return 6;
                cut_runs += ps.size() > 1;
                ++checked;
            }
        }
    }
    std::printf("piece cuts ok: %llu images, %llu of them cut\n", (unsigned long long)checked, (unsigned long long)cut_runs);
    return 0;
}
