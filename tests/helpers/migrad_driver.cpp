// CPU driver of the MIGRAD state machine (vega_amd/csrc/vmx_migrad.h), built by tests/test_migrad_machine.py with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// and run without a GPU.  It plays the role of the library's fit kernels (vmx_fit.h): every round it advances all fits,
// prints the parameter rows they ask for, and reads the function values back - the function itself lives in the test.
//
// argv[1]: capacity N of the fit state (4, 8, 16 or 32; default 32)
// stdin:  n_stages n_params iterate maxfcn up tol
//         per stage: n, then n lines "col has_lo has_hi lo hi err"
//         n_fits, then n_fits rows of n_params start values
//         then, per round, as many values as points were printed
// stdout: "ROUND <points>" + per point "<fit> <n_params values>"; at the end per stage and fit
//         "RESULT <stage> <fit> <fval> <edm> <flags> <nfcn> <n_iter>" + lines "X ...", "EXT ...", "V ..." and finally "END <rounds>"
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../vega_amd/csrc/vmx_migrad.h"

using namespace vmx_migrad;

static double read_double()
{
    char buf[64];
    if (std::scanf("%63s", buf) != 1) { std::printf("FAIL input ended\n"); std::exit(2); }
    return std::strtod(buf, nullptr);          // (accepts inf / nan)
}
static long read_int() { return (long)read_double(); }

template <int N>
static int run()
{
    Spec sp{};
    sp.n_stages = (int32_t)read_int(); sp.n_params = (int32_t)read_int(); sp.iterate = (int32_t)read_int();
    sp.maxfcn = (int32_t)read_int(); sp.up = read_double(); sp.tol = read_double();
    if (sp.n_stages < 1 || sp.n_stages > MAX_STAGES) { std::printf("FAIL stages\n"); return 2; }
    for (int s = 0; s < sp.n_stages; ++s) {
        StageSpec& st = sp.stage[s];
        st.n = (int32_t)read_int();
        if (st.n < 1 || st.n > N) { std::printf("FAIL n\n"); return 2; }
        for (int i = 0; i < st.n; ++i) {
            st.col[i] = (int32_t)read_int(); st.has_lo[i] = (int32_t)read_int(); st.has_hi[i] = (int32_t)read_int();
            st.lo[i] = read_double(); st.hi[i] = read_double(); st.err[i] = read_double();
        }
    }
    const int F = (int)read_int(), P = sp.n_params;
    std::vector<double> base((size_t)F * P);
    for (auto& v : base) v = read_double();

    std::vector<FitStateT<N>> state(F);
    for (auto& s : state) { s = FitStateT<N>{}; reset(s); }
    std::vector<std::vector<double>> ox(sp.n_stages), oext(sp.n_stages), oV(sp.n_stages), ofval(sp.n_stages), oedm(sp.n_stages);
    std::vector<std::vector<int32_t>> oflags(sp.n_stages), oiter(sp.n_stages);
    std::vector<std::vector<int64_t>> onfcn(sp.n_stages);
    StageOut outs[MAX_STAGES]{};
    for (int s = 0; s < sp.n_stages; ++s) {
        const int n = sp.stage[s].n;
        ox[s].assign((size_t)F * n, 0.); oext[s].assign((size_t)F * n, 0.); oV[s].assign((size_t)F * n * n, 0.);
        ofval[s].assign(F, 0.); oedm[s].assign(F, 0.); oflags[s].assign(F, 0); oiter[s].assign(F, 0); onfcn[s].assign(F, 0);
        outs[s] = StageOut{ox[s].data(), oext[s].data(), oV[s].data(), ofval[s].data(), oedm[s].data(), oflags[s].data(), onfcn[s].data(), oiter[s].data()};
    }
    std::vector<int> count(F, 0), offset(F + 1, 0);
    std::vector<double> vals;
    long rounds = 0;
    for (;;) {
        int total = 0;
        for (int f = 0; f < F; ++f) {
            int c = 0;
            if (!state[f].done) c = advance(state[f], sp, vals.data() + offset[f], base.data() + (size_t)f * P, outs, f);
            const int n = sp.stage[state[f].stage].n;
            if (c != request_count(state[f], n) || c > max_request(n)) { std::printf("FAIL request count\n"); return 2; }
            count[f] = c;
        }
        for (int f = 0; f < F; ++f) { offset[f] = total; total += count[f]; }
        offset[F] = total;
        if (total == 0) break;
        ++rounds;
        std::printf("ROUND %d\n", total);
        for (int f = 0; f < F; ++f) {
            const StageSpec& st = sp.stage[state[f].stage];
            double pt[MAXN];
            std::vector<double> row(P);
            for (int q = 0; q < count[f]; ++q) {
                request_point(state[f], st.n, q, pt);
                for (int i = 0; i < st.n; ++i)
                    if (request_coord(state[f], st.n, q, i) != pt[i]) { std::printf("FAIL request_coord\n"); return 2; }
                for (int c = 0; c < P; ++c) row[c] = base[(size_t)f * P + c];
                for (int i = 0; i < st.n; ++i) row[st.col[i]] = int2ext(st, i, pt[i]);
                std::printf("%d", f);
                for (int c = 0; c < P; ++c) std::printf(" %.17g", row[c]);
                std::printf("\n");
            }
        }
        std::fflush(stdout);
        vals.resize(total);
        for (auto& v : vals) v = read_double();
    }
    for (int s = 0; s < sp.n_stages; ++s) {
        const int n = sp.stage[s].n;
        for (int f = 0; f < F; ++f) {
            std::printf("RESULT %d %d %.17g %.17g %d %lld %d\nX", s, f, ofval[s][f], oedm[s][f], oflags[s][f], (long long)onfcn[s][f], oiter[s][f]);
            for (int i = 0; i < n; ++i) std::printf(" %.17g", ox[s][(size_t)f * n + i]);
            std::printf("\nEXT");
            for (int i = 0; i < n; ++i) std::printf(" %.17g", oext[s][(size_t)f * n + i]);
            std::printf("\nV");
            for (int i = 0; i < n * n; ++i) std::printf(" %.17g", oV[s][(size_t)f * n * n + i]);
            std::printf("\n");
        }
    }
    std::printf("END %ld\n", rounds);
    return 0;
}

// argv[1]: capacity of the fit state (free parameters per Minuit object) the machine is instantiated with - the library picks
// the smallest of 4 / 8 / 16 / 32 that holds the stages (vmx_fit.h)
int main(int argc, char** argv)
{
    const int cap = argc > 1 ? std::atoi(argv[1]) : MAXN;
    switch (cap) {
    case 4: return run<4>();
    case 8: return run<8>();
    case 16: return run<16>();
    case 32: return run<32>();
    default: std::printf("FAIL capacity\n"); return 2;
    }
}
