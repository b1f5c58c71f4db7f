// Host-side planning code of libvegamx: plain C++17, no HIP types, no device calls - what decides determinism and
// correctness before a kernel is launched.  Kept in its own header so that it can be compiled and run on a CPU under
// AddressSanitizer / UBSan (tests/test_planner_host.py builds tests/helpers/planner_driver.cpp with g++ -fsanitize=...).
//
//   plan_quad_tape     the persistent tape of the quadratic-form launch: cut, slot numbering, block queues
//   check_quad_tape    the invariants of a tape (coverage, balance, slot order), for the tests and for debug builds
//   choose_group_splits  K splits of the problems of a grouped product launch (two launch models)
//   csr_problem        validity / canonical form of a CSR matrix handed over the C ABI
//   cholesky_lower     in-place Cholesky factor of a symmetric positive-definite matrix (the factored chi2 form)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

// One entry of the quadratic-form tape: K range [kbeg, kend) of the 64 x 64 tile (row tile mt, walker tile nt) of problem
// `prob`; its contraction partials go to slot `slot` (k_gemm_nt44 / k_chi2_parts, vmx_device.h).
struct GemmWork { int32_t prob, mt, nt, kbeg, kend, slab, slot, pad; };

namespace vmx_plan {

struct TapeProblem { int nq, nq_pad; };       // rows of the half-form matrix and its padded leading dimension (multiple of bk)

// Row tiling of a half-form (lower-triangular) problem.  nq is rarely a multiple of the 64-row tile, and a tile costs its K
// range whatever its live rows: with the ragged tile at the BOTTOM of the triangle it is the longest row of all (5000 rows: 8
// live rows, 157 K stages - 2 % of the whole launch; 2512 rows: 16 live rows, 79 stages).  The tiles are therefore counted from
// the bottom: tile 0 holds the first `row0 = nq mod 64` rows (K range: one stage), tile t >= 1 the rows [row0 + 64 (t - 1),
// row0 + 64 t).  (Only for an even row0: the contraction epilogue fetches its operands in 16-byte pieces from the tile's first
// row on; an odd one keeps the plain tiling.)
inline int tape_row0(int nq, int bm) { const int r = nq % bm; return r % 2 == 0 ? r : 0; }
inline int tape_tile_hi(int nq, int mt, int bm)            // one past the last row a tile can hold (not clipped to nq)
{
    const int row0 = tape_row0(nq, bm);
    return row0 > 0 ? row0 + mt * bm : (mt + 1) * bm;
}
inline int tape_tile_stages(const TapeProblem& p, int mt, int bm, int bk) { return std::min((tape_tile_hi(p.nq, mt, bm) + bk - 1) / bk, p.nq_pad / bk); }

struct Tape {
    std::vector<GemmWork> work;     // every block's entries, block after block
    std::vector<int32_t> queue;     // [n_blocks + 1]: block p walks work[queue[p]] .. work[queue[p + 1] - 1]
    std::vector<int32_t> nt_off;    // [tn + 1]: slots of walker tile nt are [nt_off[nt], nt_off[nt + 1])
    std::vector<int32_t> piece_at;  // [n_pieces]: the piece (in tape order) that position xcd * per_xcd + j of the block grid takes
    int n_blocks = 0, group_size = 1, n_pieces = 0;
    int64_t n_slots = 0;
    double capacity = 0.0;          // cost bound of a piece (K stages + entry charges) the bisection ended on
};

// The quadratic-form launch as a persistent tape ("stream-K"): every (row tile, walker tile) K range of every item's
// half-triangle product is laid on ONE tape - long rows first, the walker tiles of a row next to each other - and the tape
// is cut into as many pieces of equal cost as there are resident blocks, cost = K stages + a fixed charge per entry
// (pipeline fill and contraction epilogue, `entry_stages`).  A piece boundary inside a K range splits it into two entries:
// the launch has (ranges + blocks - 1) entries at most, every block the same work to a stage, and no last round.
// The cut is a pure function of (problem shapes, walker tiles, blocks): partial sums are grouped identically on every rank
// and in every run.  Slots are numbered per walker tile in tape order; k_chi2_parts adds them in that order.
// `k_bands`: which pieces an XCD takes.  false: runs of consecutive pieces (the same or adjacent row tiles: their K segments
// tile the rows, so an XCD's blocks read every column of the walker operand at once and it streams through the 4 MB L2).
// true: the pieces are dealt to the XCDs by where they sit in K - the blocks of an XCD then sweep the same band of the walker
// operand at the same time and find it in their L2.  The pieces themselves, their cost and the slot numbering are the same
// either way: which block computes an entry changes, the sums do not (bit for bit).
inline Tape plan_quad_tape(const std::vector<TapeProblem>& probs, int tn, int blocks, double entry_stages, double skew,
                           int bm = 64, int bk = 32, bool k_bands = false)
{
    Tape T;
    const int P = std::max(blocks / 8 * 8, 8);
    struct Range { int prob, mt, stages; };
    std::vector<Range> ranges;
    for (size_t q = 0; q < probs.size(); ++q) {
        const int tm = (probs[q].nq + bm - 1) / bm;
        for (int mt = 0; mt < tm; ++mt) ranges.push_back({(int)q, mt, tape_tile_stages(probs[q], mt, bm, bk)});
    }
    std::stable_sort(ranges.begin(), ranges.end(), [](const Range& a, const Range& b) { return a.stages > b.stages; });
    // Blocks work in lock-step groups of `gs` (4 when the walker tiles allow it): the members of a group walk the SAME
    // entries, each for its own walker tile, side by side on one XCD - the matrix tile of an entry is then fetched from HBM
    // once and found in that XCD's L2 by the other members (without the lock-step every block streams its own copy: 1 GB of
    // L2 misses per launch, and the launch is bandwidth-bound).  The tape therefore carries (row tile, group of gs walker
    // tiles) ranges and is cut into P / gs pieces.
    // (... and an XCD has to hold a whole group: P / 8 blocks each)
    const int gs = (tn % 4 == 0 && P / 8 >= 4) ? 4 : (tn % 2 == 0 && P / 8 >= 2) ? 2 : 1;
    const int n_groups = tn / gs, n_pieces = 8 * (P / 8 / gs);        // (whole lock-step groups per XCD)
    const double ovh = entry_stages;
    double stages_total = 0.0;
    for (auto& r : ranges) stages_total += (double)r.stages * n_groups;
    constexpr int MIN_SEG = 2;                  // no entry shorter than this many stages (but for ranges that short)
    struct Entry { GemmWork w; int piece; };    // (w.nt: the group's first walker tile)
    std::vector<Entry> entries;
    // A piece may cost at most `cap` = its K stages + the fixed charge of each of its entries; the tape is filled greedily -
    // a piece takes what fits, a cut inside a K range opens an entry on either side - and `cap` is the smallest capacity for
    // which n_pieces pieces suffice (bisection: the piece count is monotone in cap).  Every piece then costs cap to within a
    // stage, but for the last one, which holds what is left.
    // (a CU issues from its older resident block first: of two equal pieces the one dispatched first ends earlier and leaves the
    // other alone on the CU - block p = 8 i + xcd takes piece xcd * per_xcd + i / gs, so the first half of an XCD's pieces belongs
    // to the blocks dispatched first; `skew` shifts work to them)
    const int per_xcd_pieces = std::max(n_pieces / 8, 1);
    // (the same tape whatever the number of batches in flight - a walker's chi2 must not depend on it, and the cut points decide
    // how its partial sums are grouped)
    auto weight = [&](int pc) { return (pc % per_xcd_pieces) < per_xcd_pieces / 2 ? 1.0 + skew : 1.0 - skew; };
    auto fill = [&](double cap0, bool keep) {
        if (keep) entries.clear();
        int pc = 0;
        double cur = 0.0, cap = cap0 * weight(0);
        for (auto& r : ranges)
            for (int grp = 0; grp < n_groups; ++grp) {
                int k = 0, left = r.stages;
                while (left > 0) {
                    const double avail = cap - cur - ovh;
                    if (cur > 0.0 && avail < (double)std::min(left, MIN_SEG)) { ++pc; cur = 0.0; cap = cap0 * weight(std::min(pc, n_pieces - 1)); continue; }
                    int take = std::min(left, std::max(MIN_SEG, (int)std::floor(avail + 1e-9)));
                    const int rem = left - take;
                    if (rem > 0 && rem < MIN_SEG) take = (take - (MIN_SEG - rem) >= MIN_SEG) ? take - (MIN_SEG - rem) : left;     // no sliver behind the cut
                    if (keep) entries.push_back({GemmWork{r.prob, r.mt, grp * gs, k * bk, (k + take) * bk, 0, 0, 0}, std::min(pc, n_pieces - 1)});
                    cur += take + ovh;
                    k += take; left -= take;
                }
            }
        return pc + 1;
    };
    {
        double lo = stages_total / n_pieces, hi = lo + ovh * (double)(ranges.size() * n_groups) / n_pieces + 4.0 * ovh + 256.0;
        // (an odd number of pieces per XCD: the weights no longer average to one, the upper bound has to allow for the lighter half)
        if (per_xcd_pieces % 2) { lo /= 1.0 + std::fabs(skew); hi /= 1.0 - std::fabs(skew); }
        for (int it = 0; it < 48; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (fill(mid, false) <= n_pieces) hi = mid; else lo = mid;
        }
        (void)fill(hi, true);
        T.capacity = hi;
    }
    // slots: per walker tile, tape order (entry j of walker-tile group grp is slot (its rank within the group) of every member)
    std::vector<int32_t> nt_off(tn + 1, 0);
    for (auto& en : entries)
        for (int m = 0; m < gs; ++m) ++nt_off[en.w.nt + m + 1];
    for (int nt = 0; nt < tn; ++nt) nt_off[nt + 1] += nt_off[nt];
    {
        std::vector<int32_t> next(n_groups, 0);
        for (auto& en : entries) en.w.slot = next[en.w.nt / gs]++;          // (rank within the group: + nt_off[nt] per member)
    }
    // queues: block p = 8 i + xcd is member i % gs of the group that takes piece xcd * (P / 8 / gs) + i / gs: the members of a
    // group are neighbouring blocks of one XCD (the dispatcher hands them to one or two CUs), start together and stay close
    // - 496 MB of L2 misses per launch at B = 256 and 149.7 us.  (Spreading a group's members over the XCD's CUs, so that
    // co-resident blocks are out of phase, was measured at 616 MB and 153.4 us: the members drift apart and the matrix tile
    // is fetched again.)
    std::vector<int32_t> queue(P + 1, 0);
    std::vector<std::vector<GemmWork>> by_piece(n_pieces);
    for (auto& en : entries) by_piece[en.piece].push_back(en.w);
    // which piece a grid position takes.  A position's weight (the first half of an XCD's positions belongs to the blocks
    // dispatched first and carries 1 + skew) must be its piece's: the heavy and the light pieces are dealt separately.
    std::vector<int32_t> piece_at(n_pieces);
    for (int pc = 0; pc < n_pieces; ++pc) piece_at[pc] = pc;
    if (k_bands && n_pieces >= 16) {
        // a piece's place in K: the stage-weighted mean of its entries' K midpoints
        std::vector<double> where(n_pieces, 0.0);
        for (int pc = 0; pc < n_pieces; ++pc) {
            double wsum = 0.0, ksum = 0.0;
            for (auto& w : by_piece[pc]) { const double len = w.kend - w.kbeg; wsum += len; ksum += len * 0.5 * (w.kbeg + w.kend); }
            where[pc] = wsum > 0.0 ? ksum / wsum : 1e300;          // (empty pieces last)
        }
        const int half = per_xcd_pieces / 2;
        std::vector<int32_t> heavy, light;
        for (int pc = 0; pc < n_pieces; ++pc) ((pc % per_xcd_pieces) < half ? heavy : light).push_back(pc);
        auto by_k = [&](int32_t a, int32_t b) { return where[a] < where[b]; };
        std::stable_sort(heavy.begin(), heavy.end(), by_k);
        std::stable_sort(light.begin(), light.end(), by_k);
        for (int x = 0; x < 8; ++x)
            for (int j = 0; j < per_xcd_pieces; ++j)
                piece_at[x * per_xcd_pieces + j] = j < half ? heavy[(size_t)x * half + j] : light[(size_t)x * (per_xcd_pieces - half) + (j - half)];
    }
    T.work.reserve(entries.size() * gs);
    for (int p = 0; p < P; ++p) {
        const int xcd = p % 8, i = p / 8, per_xcd = P / 8 / gs;
        const int member = i % gs;
        const int pos = xcd * per_xcd + i / gs;
        queue[p] = (int32_t)T.work.size();
        if (pos < n_pieces && i / gs < per_xcd)
            for (auto w : by_piece[piece_at[pos]]) {
                w.nt += member;
                w.slot += nt_off[w.nt];
                T.work.push_back(w);
            }
    }
    queue[P] = (int32_t)T.work.size();
    T.queue = std::move(queue);
    T.n_slots = nt_off[tn];
    T.nt_off = std::move(nt_off);
    T.n_blocks = P; T.group_size = gs; T.n_pieces = n_pieces;
    T.piece_at = std::move(piece_at);
    return T;
}

// Invariants of a tape; returns "" or what is wrong.  (i) every (problem, row tile, walker tile) K range is covered exactly
// once by contiguous, non-empty, stage-aligned segments; (ii) every slot of every walker tile is written by exactly one entry,
// and within a walker tile the slots follow tape order (row tiles by decreasing length, K ascending); (iii) the pieces' costs
// (stages + entry charges) stay within one entry (charge + MIN_SEG stages) of the capacity; (iv) the members of a lock-step
// group carry the same entries for consecutive walker tiles.
inline std::string check_quad_tape(const Tape& T, const std::vector<TapeProblem>& probs, int tn, double entry_stages, double skew,
                                   int bm = 64, int bk = 32)
{
    if ((int)T.queue.size() != T.n_blocks + 1 || (int)T.nt_off.size() != tn + 1) return "queue / nt_off sizes";
    if (T.queue[0] != 0 || T.queue[T.n_blocks] != (int32_t)T.work.size()) return "queue ends";
    for (int p = 0; p < T.n_blocks; ++p) if (T.queue[p + 1] < T.queue[p]) return "queue not monotone";
    // (i) coverage
    std::vector<std::vector<std::vector<std::pair<int, int>>>> seg(probs.size());
    for (size_t q = 0; q < probs.size(); ++q) seg[q].resize((size_t)((probs[q].nq + bm - 1) / bm) * tn);
    std::vector<char> slot_used((size_t)T.n_slots, 0);
    for (auto& w : T.work) {
        if (w.prob < 0 || w.prob >= (int)probs.size()) return "problem index";
        const int tm = (probs[w.prob].nq + bm - 1) / bm;
        if (w.mt < 0 || w.mt >= tm || w.nt < 0 || w.nt >= tn) return "tile index";
        if (w.kbeg % bk || w.kend % bk || w.kend <= w.kbeg) return "segment not stage-aligned or empty";
        if (w.slot < T.nt_off[w.nt] || w.slot >= T.nt_off[w.nt + 1]) return "slot outside its walker tile's range";
        if (slot_used[w.slot]++) return "slot written twice";
        seg[w.prob][(size_t)w.mt * tn + w.nt].push_back({w.kbeg, w.kend});
    }
    for (auto u : slot_used) if (!u) return "slot never written";
    for (size_t q = 0; q < probs.size(); ++q) {
        const int tm = (probs[q].nq + bm - 1) / bm;
        for (int mt = 0; mt < tm; ++mt)
            for (int nt = 0; nt < tn; ++nt) {
                auto v = seg[q][(size_t)mt * tn + nt];
                std::sort(v.begin(), v.end());
                const int want = tape_tile_stages(probs[q], mt, bm, bk) * bk;
                int at = 0;
                for (auto& s : v) { if (s.first != at) return "K range has a gap or an overlap"; at = s.second; }
                if (at != want) return "K range not covered to its end";
            }
    }
    // (ii) slots of a walker tile in tape order: sort its entries by slot; (stages desc, prob, mt) must not increase in length
    for (int nt = 0; nt < tn; ++nt) {
        std::vector<const GemmWork*> mine;
        for (auto& w : T.work) if (w.nt == nt) mine.push_back(&w);
        std::sort(mine.begin(), mine.end(), [](const GemmWork* a, const GemmWork* b) { return a->slot < b->slot; });
        for (size_t j = 1; j < mine.size(); ++j) {
            const GemmWork &a = *mine[j - 1], &b = *mine[j];
            if (a.prob == b.prob && a.mt == b.mt) { if (b.kbeg != a.kend) return "segments of a K range out of slot order"; continue; }
            const int la = tape_tile_stages(probs[a.prob], a.mt, bm, bk);
            const int lb = tape_tile_stages(probs[b.prob], b.mt, bm, bk);
            if (lb > la) return "row tiles out of tape order (long rows first)";
            if (b.kbeg != 0) return "a K range starts in the middle";
        }
    }
    // (iii) balance and (iv) lock-step
    const int gs = T.group_size, per_xcd = T.n_blocks / 8 / gs, per_xcd_pieces = std::max(T.n_pieces / 8, 1);
    for (int p = 0; p < T.n_blocks; ++p) {
        const int xcd = p % 8, i = p / 8, pos = xcd * per_xcd + i / gs;
        if (i / gs >= per_xcd) { if (T.queue[p + 1] != T.queue[p]) return "a surplus block has work"; continue; }
        const int pcs = T.piece_at[pos];
        if ((pcs % per_xcd_pieces < per_xcd_pieces / 2) != (pos % per_xcd_pieces < per_xcd_pieces / 2)) return "a piece sits on a position of the other weight class";
        double cost = 0.0;
        for (int j = T.queue[p]; j < T.queue[p + 1]; ++j) cost += (T.work[j].kend - T.work[j].kbeg) / bk + entry_stages;
        const double w = (pcs % per_xcd_pieces) < per_xcd_pieces / 2 ? 1.0 + skew : 1.0 - skew;
        // (a piece takes at least one entry of MIN_SEG = 2 stages whatever its capacity, and a cut leaves no sliver shorter than
        // that behind it: one entry charge + 2 MIN_SEG stages of slack)
        if (cost > T.capacity * w + entry_stages + 4.0 + 1e-6) return "a piece exceeds its capacity by more than one entry";
        if (pcs + 1 < T.n_pieces && cost > 0.0 && cost < T.capacity * w - entry_stages - 4.0) {
            // (only the last non-empty piece may be short)
            bool later = false;
            for (int p2 = 0; p2 < T.n_blocks && !later; ++p2) {
                if ((p2 / 8) / gs >= per_xcd) continue;
                const int pcs2 = T.piece_at[(p2 % 8) * per_xcd + (p2 / 8) / gs];
                if (pcs2 > pcs && T.queue[p2 + 1] > T.queue[p2]) later = true;
            }
            if (later) return "a piece before the last is short";
        }
        if (i % gs != 0) {
            const int lead = p - 8 * (i % gs);
            if (T.queue[p + 1] - T.queue[p] != T.queue[lead + 1] - T.queue[lead]) return "lock-step group members differ in length";
            for (int j = 0; j < T.queue[p + 1] - T.queue[p]; ++j) {
                const GemmWork &a = T.work[T.queue[lead] + j], &b = T.work[T.queue[p] + j];
                if (a.prob != b.prob || a.mt != b.mt || a.kbeg != b.kbeg || a.kend != b.kend || b.nt != a.nt + i % gs) return "lock-step group members differ";
            }
        }
    }
    return "";
}

// K splits of the problems of one grouped product launch (the distortion products of all correlation items, the factored chi2
// form's products): every combination of 1 / 2 / 4 / 8-way splits is simulated (up to three problems) and the shortest makespan
// wins, extra slabs charged with their write + re-read at ~3 TB/s.  Two models of the launch:
//   one block per CU (two_per_cu = false): blocks of `stages / split` K stages (+ a fixed start / end cost) handed in launch
//     order to the first free of 256 CUs - a CU works through short blocks one after the other (the SIMDs issue from the oldest
//     wave first); fitted on the full chain's distortion products (round 2);
//   two half-speed blocks per CU, XCD by XCD (two_per_cu = true): two resident blocks share a CU's MFMA pipes for as long as
//     both are there, and a block that is alone on its CU does NOT run twice as fast (0.55 - 0.65 of the pair's rate: it cannot
//     cover its own barriers and load latencies); and a K split belongs to whole XCDs (XCD x works on split x % nsplit of the
//     row tiles of its group, k_gemm_nt44), so an XCD's load is what counts.  For launches of few long blocks this is the
//     better model: the factored chi2 form at COEFMOD = 2, B = 256 (300 tiles of 313 / 625 stages) - measured 0.976 ms
//     unsplit, 0.983 (2, 1), 0.806 (2, 2), 0.646 (4, 2), 0.659 (4, 4); this model: 1258, 1273, 964, 662, 677 (round 4).
struct SplitProblem { int tiles; int stages; int64_t slab_bytes; int max_split; int tm; int tn; };
inline std::vector<int> choose_group_splits(const std::vector<SplitProblem>& probs, bool two_per_cu = false)
{
    const int n = (int)probs.size();
    std::vector<int> best(n, 0);
    if (n == 0 || n > 3) return best;
    constexpr double BLOCK_OVERHEAD = 4.0;                  // pipeline fill + epilogue of a block, in K stages
    constexpr double STAGE_US = 0.85, SLAB_BYTES_PER_US = 3.0e6;
    double best_cost = 1e300;
    std::vector<int> cur(n, 1);
    const int combos = 1 << (2 * n);
    auto schedule = [](std::vector<double>& free_at, double cost, int blocks, double& makespan) {
        // list scheduling: every block goes to the slot that falls free first (free_at is kept as a min-heap)
        for (int b = 0; b < blocks; ++b) {
            std::pop_heap(free_at.begin(), free_at.end(), std::greater<double>());
            const double t = free_at.back() + cost;
            free_at.back() = t;
            std::push_heap(free_at.begin(), free_at.end(), std::greater<double>());
            if (t > makespan) makespan = t;
        }
    };
    for (int c = 0; c < combos; ++c) {
        bool ok = true;
        double penalty = 0.0;
        for (int i = 0; i < n; ++i) {
            cur[i] = 1 << ((c >> (2 * i)) & 3);
            if (cur[i] > probs[i].max_split || probs[i].stages / cur[i] < 4) ok = false;
            penalty += (cur[i] - 1) * 2.0 * (double)probs[i].slab_bytes / SLAB_BYTES_PER_US / STAGE_US;
        }
        if (!ok) continue;
        double makespan = 0.0;
        if (!two_per_cu) {
            std::vector<double> free_at(256, 0.0);
            for (int i = 0; i < n; ++i)
                schedule(free_at, (double)((probs[i].stages + cur[i] - 1) / cur[i]) + BLOCK_OVERHEAD, probs[i].tiles * cur[i], makespan);
        } else {
            for (int x = 0; x < 8; ++x) {
                std::vector<double> free_at(64, 0.0);           // 32 CUs x 2 resident blocks
                for (int i = 0; i < n; ++i) {
                    const int ngroups = 8 / cur[i], group = x / cur[i];
                    const int rows = probs[i].tm > group ? (probs[i].tm - group + ngroups - 1) / ngroups : 0;      // row tiles group, group + ngroups, ...
                    schedule(free_at, 2.0 * ((double)((probs[i].stages + cur[i] - 1) / cur[i]) + BLOCK_OVERHEAD), rows * probs[i].tn, makespan);
                }
            }
        }
        const double total = makespan + penalty;
        if (total < best_cost) { best_cost = total; best = cur; }
    }
    return best;
}

// A CSR matrix handed over the C ABI (vmx_item_set_matrix_csr): "" or what is wrong.  Canonical form is required: the set-up
// of the quadratic form scatters the rows (last write wins) where the product adds them, so duplicates would disagree.
inline const char* csr_problem(int32_t rows, int32_t cols, const int64_t* indptr, const int32_t* indices)
{
    if (rows < 0 || cols < 0 || !indptr) return "shape";
    if (indptr[0] != 0) return "indptr[0] must be 0";
    for (int r = 0; r < rows; ++r) if (indptr[r + 1] < indptr[r]) return "indptr must be non-decreasing";
    const int64_t nnz = indptr[rows];
    if (nnz > 0 && !indices) return "indices";
    for (int64_t k = 0; k < nnz; ++k) if (indices[k] < 0 || indices[k] >= cols) return "column index out of range";
    for (int r = 0; r < rows; ++r)
        for (int64_t k = indptr[r] + 1; k < indptr[r + 1]; ++k)
            if (indices[k] <= indices[k - 1]) return "column indices must be strictly ascending within a row (no duplicates)";
    return "";
}

// In-place Cholesky factor of the symmetric positive-definite matrix a [n][ld] (lower triangle read): on return the lower
// triangle holds L with a = L L^T, the strict upper triangle is zeroed.  Row-oriented (Cholesky-Banachiewicz): the inner
// loops are contiguous dot products.  Returns false when a pivot is not positive (the matrix is left partly factored).
inline bool cholesky_lower(double* a, int n, int64_t ld)
{
    for (int i = 0; i < n; ++i) {
        double* ri = a + (int64_t)i * ld;
        for (int j = 0; j <= i; ++j) {
            const double* rj = a + (int64_t)j * ld;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = 0;
            for (; k + 3 < j; k += 4) { s0 += ri[k] * rj[k]; s1 += ri[k + 1] * rj[k + 1]; s2 += ri[k + 2] * rj[k + 2]; s3 += ri[k + 3] * rj[k + 3]; }
            for (; k < j; ++k) s0 += ri[k] * rj[k];
            const double v = ri[j] - ((s0 + s1) + (s2 + s3));
            if (i == j) {
                if (!(v > 0.0)) return false;
                ri[j] = std::sqrt(v);
            } else ri[j] = v / rj[j];
        }
        for (int j = i + 1; j < n; ++j) ri[j] = 0.0;
    }
    return true;
}

}  // namespace vmx_plan
