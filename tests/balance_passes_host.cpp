// Host restatement of the flatten step's balancing passes (spx_flatten.hip: flatten_on_device) around the SAME cut rule
// (spx_layout.h: for_each_piece): rows after every pass and the longest image before it, for a run list read from a file
// (u64 r, r head bytes, r u64 lengths).  tests/test_piece_cuts.py holds its figures against the ones the device passes
// printed for the same run list (profiles/r03_balanced_pieces_default_build.txt).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "../spumoni_amd/csrc/spx_layout.h"

using namespace spx;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const uint32_t span = (uint32_t)std::atoi(argv[2]);
    const int passes = std::atoi(argv[3]);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t r = 0;
    if (std::fread(&r, 8, 1, f) != 1) return 2;
    std::vector<uint8_t> heads(r);
    std::vector<uint64_t> lens(r);
    if (std::fread(heads.data(), 1, r, f) != r || std::fread(lens.data(), 8, r, f) != r) return 2;
    std::fclose(f);
    for (int pass = 0; pass < passes; ++pass) {
        r = heads.size();
        std::vector<uint64_t> S(r + 1, 0);
        for (uint64_t k = 0; k < r; ++k) S[k + 1] = S[k] + lens[k];
        std::vector<uint32_t> order(r);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return std::max<uint8_t>(heads[x], 1) < std::max<uint8_t>(heads[y], 1);
        });
        std::vector<uint64_t> lf(r);
        uint64_t acc = 0;
        for (uint32_t k : order) {
            lf[k] = acc;
            acc += lens[k];
        }
        std::vector<uint8_t> h2;
        std::vector<uint64_t> l2;
        uint64_t span_max = 0;
        for (uint64_t k = 0; k < r; ++k) {
            const uint64_t a = run_of_position(S.data(), r, lf[k]);
            const uint64_t nb = run_of_position(S.data(), r, lf[k] + lens[k] - 1) - a;
            if (nb + 1 > span) span_max = std::max(span_max, nb + 1);
            for_each_piece(lens[k], lf[k], S.data(), a, nb, span, [&](uint64_t, uint64_t plen) {
                h2.push_back(heads[k]);
                l2.push_back(plen);
            });
        }
        std::printf("pass %d: %llu rows -> %llu; longest image %llu\n", pass, (unsigned long long)r, (unsigned long long)h2.size(),
                    (unsigned long long)span_max);
        if (h2.size() == r) break;
        heads.swap(h2);
        lens.swap(l2);
    }
    return 0;
}
