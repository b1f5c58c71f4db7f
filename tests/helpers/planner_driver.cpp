// CPU driver of the library's host-side planners (vega_amd/csrc/vmx_plan.h), built by tests/test_planner_host.py with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// and run without a GPU: prints one line per case, "FAIL ..." and a non-zero exit code when an invariant breaks.
#include <cstdio>
#include <cstring>
#include <random>

#include "../../vega_amd/csrc/vmx_plan.h"

using namespace vmx_plan;

static int failures = 0;
static void expect(bool ok, const char* what)
{
    if (!ok) { std::printf("FAIL %s\n", what); ++failures; }
}

static int pad32(int n) { return (n + 31) / 32 * 32; }

static void tape_cases()
{
    const std::vector<std::vector<int>> shapes = {
        {2512, 5000},                // the bench's joint fit (auto + cross, 12 broadband coefficients on the auto)
        {2512},                      // auto only
        {10012, 20000},              // COEFMOD = 2 model grids
        {70}, {64}, {65, 1}, {3180, 1590, 2500, 5000},
    };
    const int batches[] = {9, 64, 256, 512, 4096};
    const int block_counts[] = {512, 608, 200, 8};       // 2 x 256 CUs (MI355X), 2 x 304, a count that is no multiple of 32, the minimum
    for (auto& sh : shapes)
        for (int B : batches)
            for (int blocks : block_counts) {
                std::vector<TapeProblem> probs;
                for (int n : sh) probs.push_back({n, pad32(n)});
                const int tn = (B + 63) / 64;
                const Tape a = plan_quad_tape(probs, tn, blocks, 4.0, 0.12);
                const Tape b = plan_quad_tape(probs, tn, blocks, 4.0, 0.12);
                const std::string why = check_quad_tape(a, probs, tn, 4.0, 0.12);
                char label[160];
                std::snprintf(label, sizeof label, "tape nq0=%d items=%zu B=%d blocks=%d: %s", sh[0], sh.size(), B, blocks, why.c_str());
                expect(why.empty(), label);
                // identical inputs -> identical plan, byte for byte
                const bool same = a.work.size() == b.work.size() && a.queue == b.queue && a.nt_off == b.nt_off &&
                                  (a.work.empty() || std::memcmp(a.work.data(), b.work.data(), a.work.size() * sizeof(GemmWork)) == 0);
                expect(same, "identical inputs give identical tapes");
                // K bands per XCD: the same pieces on other blocks - every invariant holds, and entry for entry the same
                // (problem, tile, K range) -> slot map, i.e. the same sums bit for bit
                {
                    const Tape k = plan_quad_tape(probs, tn, blocks, 4.0, 0.12, 64, 32, true);
                    const std::string whyk = check_quad_tape(k, probs, tn, 4.0, 0.12);
                    std::snprintf(label, sizeof label, "banded tape nq0=%d items=%zu B=%d blocks=%d: %s", sh[0], sh.size(), B, blocks, whyk.c_str());
                    expect(whyk.empty(), label);
                    auto key = [](const GemmWork& w) { return std::vector<int32_t>{w.prob, w.mt, w.nt, w.kbeg, w.kend, w.slot}; };
                    std::vector<std::vector<int32_t>> ka, kk;
                    for (auto& w : a.work) ka.push_back(key(w));
                    for (auto& w : k.work) kk.push_back(key(w));
                    std::sort(ka.begin(), ka.end()); std::sort(kk.begin(), kk.end());
                    expect(ka == kk && a.nt_off == k.nt_off, "the banded tape carries the same entries with the same slots");
                }
                // the plan depends on B through the number of walker tiles only
                if (B % 64) {
                    const Tape c = plan_quad_tape(probs, tn, blocks, 4.0, 0.12);
                    expect(c.queue == a.queue, "a tape is a function of the walker-tile count");
                }
                // pieces of equal cost: the busiest and the idlest non-empty block within one entry + one skew step
                double lo = 1e300, hi = 0.0;
                int64_t stages = 0;
                for (int p = 0; p < a.n_blocks; ++p) {
                    double cost = 0.0;
                    for (int j = a.queue[p]; j < a.queue[p + 1]; ++j) { cost += (a.work[j].kend - a.work[j].kbeg) / 32 + 4.0; stages += (a.work[j].kend - a.work[j].kbeg) / 32; }
                    if (cost > 0.0) { lo = std::min(lo, cost); hi = std::max(hi, cost); }
                }
                int64_t want = 0;
                for (auto& pr : probs)
                    for (int mt = 0; mt < (pr.nq + 63) / 64; ++mt) want += (int64_t)tape_tile_stages(pr, mt, 64, 32) * tn;
                expect(stages == want, "the tape carries every K stage exactly once");
                std::printf("ok tape nq0=%d items=%zu B=%d blocks=%d: %zu entries, %lld slots, piece cost %.0f..%.0f (cap %.1f)\n",
                            sh[0], sh.size(), B, blocks, a.work.size(), (long long)a.n_slots, lo, hi, a.capacity);
            }
}

static void tiling_cases()
{
    // tiles counted from the bottom of the triangle: the ragged tile is the SHORT one
    expect(tape_row0(5000, 64) == 8 && tape_row0(2512, 64) == 16 && tape_row0(2560, 64) == 0 && tape_row0(2501, 64) == 0, "row0: nq mod 64 when even");
    const TapeProblem cross{5000, 5024};
    expect(tape_tile_stages(cross, 0, 64, 32) == 1 && tape_tile_stages(cross, 1, 64, 32) == 3 && tape_tile_stages(cross, 78, 64, 32) == 157, "stages of the offset tiling");
    for (int nq : {5000, 2512, 2500, 70, 64, 65, 20000}) {
        const TapeProblem p{nq, pad32(nq)};
        const int tm = (nq + 63) / 64, row0 = tape_row0(nq, 64);
        int covered = 0;
        for (int mt = 0; mt < tm; ++mt) {
            const int lo = row0 > 0 ? (mt == 0 ? 0 : row0 + (mt - 1) * 64) : mt * 64, hi = std::min(tape_tile_hi(nq, mt, 64), nq);
            expect(lo == covered && hi > lo && hi - lo <= 64, "tiles partition the rows");
            expect(tape_tile_stages(p, mt, 64, 32) * 32 >= hi || tape_tile_stages(p, mt, 64, 32) == p.nq_pad / 32, "a tile's K range reaches its diagonal");
            covered = hi;
        }
        expect(covered == nq, "every row in exactly one tile");
    }
    std::printf("ok tiling\n");
}

static void split_cases()
{
    // the factored chi2 form at COEFMOD = 2 (fitted bins 3180 and 1590 -> 50 and 25 row tiles; 625 and 313 K stages): the
    // per-XCD model picks the measured best (4, 2) at B = 256, the one-block-per-CU model leaves it unsplit
    auto probs = [](int B, int max_split) {
        const int tn = (B + 63) / 64;
        return std::vector<SplitProblem>{{50 * tn, 625, (int64_t)B * 3200 * 8, max_split, 50, tn}, {25 * tn, 313, (int64_t)B * 1600 * 8, max_split, 25, tn}};
    };
    expect(choose_group_splits(probs(256, 4), true) == std::vector<int>({4, 2}), "per-XCD model: (4, 2) at B = 256");
    expect(choose_group_splits(probs(256, 4), false) == std::vector<int>({1, 1}), "one-block model: unsplit at B = 256");
    expect(choose_group_splits(probs(64, 8), true) == std::vector<int>({8, 4}), "per-XCD model: (8, 4) at B = 64");
    expect(choose_group_splits(probs(1024, 1), true) == std::vector<int>({1, 1}), "no slab room: unsplit");
    expect(choose_group_splits(probs(256, 4), true) == choose_group_splits(probs(256, 4), true), "deterministic");
    expect(choose_group_splits({}, true).empty(), "no problems, no splits");
    for (int B : {9, 64, 256, 512, 4096})
        for (bool two : {false, true})
            for (int v : choose_group_splits(probs(B, 4), two)) expect(v == 1 || v == 2 || v == 4, "a split within its bound");
    std::printf("ok split chooser\n");
}

static void csr_cases()
{
    const int64_t ptr_ok[] = {0, 2, 2, 5};
    const int32_t idx_ok[] = {0, 3, 1, 2, 3};
    expect(csr_problem(3, 4, ptr_ok, idx_ok)[0] == 0, "a canonical CSR matrix is accepted");
    const int64_t ptr_bad0[] = {1, 2, 2, 5};
    expect(csr_problem(3, 4, ptr_bad0, idx_ok)[0] != 0, "indptr[0] != 0 is refused");
    const int64_t ptr_dec[] = {0, 3, 2, 5};
    expect(csr_problem(3, 4, ptr_dec, idx_ok)[0] != 0, "decreasing indptr is refused");
    const int32_t idx_dup[] = {0, 0, 1, 2, 3};
    expect(csr_problem(3, 4, ptr_ok, idx_dup)[0] != 0, "duplicate columns are refused");
    const int32_t idx_unsorted[] = {3, 0, 1, 2, 3};
    expect(csr_problem(3, 4, ptr_ok, idx_unsorted)[0] != 0, "unsorted columns are refused");
    const int32_t idx_range[] = {0, 4, 1, 2, 3};
    expect(csr_problem(3, 4, ptr_ok, idx_range)[0] != 0, "a column index out of range is refused");
    const int64_t ptr_empty[] = {0, 0, 0, 0};
    expect(csr_problem(3, 4, ptr_empty, nullptr)[0] == 0, "an empty matrix is accepted");
    std::printf("ok csr checks\n");
}

static void cholesky_cases()
{
    std::mt19937_64 rng(7);
    std::normal_distribution<double> g(0.0, 1.0);
    for (int n : {1, 2, 5, 33, 130}) {
        const int ld = pad32(n);
        std::vector<double> m((size_t)n * n), a((size_t)n * ld, 0.0), ref;
        for (auto& v : m) v = g(rng);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = i == j ? 0.5 * n : 0.0;
                for (int k = 0; k < n; ++k) s += m[(size_t)i * n + k] * m[(size_t)j * n + k];
                a[(size_t)i * ld + j] = s;
            }
        ref = a;
        expect(cholesky_lower(a.data(), n, ld), "a positive-definite matrix factors");
        double err = 0.0, scale = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += a[(size_t)i * ld + k] * a[(size_t)j * ld + k];
                err = std::max(err, std::fabs(s - ref[(size_t)i * ld + j]));
                scale = std::max(scale, std::fabs(ref[(size_t)i * ld + j]));
                if (j > i) expect(a[(size_t)i * ld + j] == 0.0, "the strict upper triangle is zeroed");
            }
        expect(err <= 1e-13 * scale * n, "L L^T reproduces the matrix");
        std::printf("ok cholesky n=%d: |L L^T - A| = %.2e of %.2e\n", n, err, scale);
    }
    std::vector<double> bad = {1.0, 2.0, 2.0, 1.0};      // indefinite
    expect(!cholesky_lower(bad.data(), 2, 2), "an indefinite matrix is reported");
}

int main()
{
    tape_cases();
    tiling_cases();
    split_cases();
    csr_cases();
    cholesky_cases();
    std::printf(failures ? "FAILED: %d\n" : "all planner checks passed\n", failures);
    return failures ? 1 : 0;
}
