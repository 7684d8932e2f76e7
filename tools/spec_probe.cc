// spec_probe.cc -- measurement tool (NOT product, NOT a test): how often could the greedy loop know its next pick before the
// table update of the current step has run?  Built on the CPU oracle's own state (it #includes oracle/cmvm_oracle.cc), method wmc.
//
//   g++ -O2 -std=c++20 -o /tmp/spec_probe tools/spec_probe.cc && /tmp/spec_probe 128 0
//
// Per step t (table T_t, pick P_t = (A, B), new row N):
//   R_t  = best entry of T_t touching neither A nor B                      (unchanged by step t)
//   V_t  = best entry of T_t touching exactly one of A / B                  (old values; some of them change in step t)
//   M_t  = best entry of T_{t+1} touching A, B or N
// The pick of step t+1 is max(R_t, M_t).  Counted:
//   disjoint   P_{t+1} == R_t                                              (exact condition: R_t > M_t)
//   bw         the same decided on the 64-bit bound word (rank << 32 | tie >> 23): strictly greater
//   oldvals    R_t > V_t and R_t > best entry of T_{t+1} touching N or being one of the three special blocks -- what a search that
//              only knows the OLD values of the blocks touching A / B can certify
#include "../oracle/cmvm_oracle.cc"

#include <random>

using namespace orc;

struct Cand {
    int64_t score = -1;
    PairKey key = NO_PAIR;
    bool valid() const { return score >= 0; }
    bool beats(const Cand &o) const {  // later in the table wins a tie
        if (!valid()) return false;
        if (!o.valid()) return true;
        if (score != o.score) return score > o.score;
        return o.key < key;
    }
    void offer(int64_t s, const PairKey &k) {
        Cand c;
        c.score = s;
        c.key = k;
        if (c.beats(*this)) *this = c;
    }
    uint64_t bw(int n_bits) const {
        if (!valid()) return 0;
        const int idx = (key.sub ? 1 : 0) * (2 * n_bits - 1) + key.shift + (n_bits - 1);
        const uint64_t tie = ((uint64_t)key.id1 << 31) | ((uint64_t)key.id0 << 7) | (uint64_t)idx;
        return ((uint64_t)(score + 1) << 32) | (tie >> 23);
    }
};

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64;
    const int seed = argc > 2 ? atoi(argv[2]) : 0;
    const long max_steps = argc > 3 ? atol(argv[3]) : (1l << 60);
    std::mt19937_64 rng(seed);
    std::vector<float> kernel((size_t)n * n);
    for (auto &v : kernel) v = (float)((int)(rng() % 255) - 127);
    std::vector<QInt> qints(n, QInt{-128.0f, 127.0f, 1.0f});
    std::vector<float> lats(n, 0.0f);
    State s = create_state(kernel.data(), n, n, qints, lats, false);
    auto score_of = [&](const Entry &e) {
        int8_t ov = overlap_accum(s.ops[e.first.id0].q, s.ops[e.first.id1].q).first;
        return (int64_t)e.second * ov;
    };
    long steps = 0, disjoint = 0, bwok = 0, oldvals = 0, run = 0, runs = 0, runsum = 0;
    long q_steps[4] = {0, 0, 0, 0}, q_dis[4] = {0, 0, 0, 0}, q_old[4] = {0, 0, 0, 0};
    std::vector<char> rec_dis, rec_old;
    Cand prevR, prevV;
    long prevL = 0;
    std::vector<long> hist_c(65, 0), hist_l(65, 0);
    int64_t pA = -1, pB = -1, pN = -1;
    while (!s.table.empty() && steps < max_steps) {
        Cand best;
        for (const auto &e : s.table) best.offer(score_of(e), e.first);
        if (!best.valid()) break;
        if (pA >= 0) {  // classify the previous step with this table = T_{t+1}
            Cand M, Mnew;
            for (const auto &e : s.table) {
                const PairKey &k = e.first;
                const bool tA = k.id0 == pA || k.id1 == pA, tB = k.id0 == pB || k.id1 == pB, tN = k.id0 == pN || k.id1 == pN;
                if (tA || tB || tN) M.offer(score_of(e), k);
                const bool special = (k.id0 == pA || k.id0 == pB) && (k.id1 == pA || k.id1 == pB);
                if (tN || special) Mnew.offer(score_of(e), k);
            }
            {  // sizes of the candidate lists an exact scheme needs: entries of T_{t+1} touching A / B / N whose bound word reaches R_t's
                long cn = 0;
                const uint64_t r0 = prevR.bw(s.n_bits);
                if (r0)
                    for (const auto &e : s.table) {
                        const PairKey &k = e.first;
                        if (!(k.id0 == pA || k.id1 == pA || k.id0 == pB || k.id1 == pB || k.id0 == pN || k.id1 == pN)) continue;
                        // only the best entry of a block (row pair) counts: the engine folds one entry per block
                        Cand c;
                        c.offer(score_of(e), k);
                        if (c.valid() && c.bw(s.n_bits) >= r0) ++cn;
                    }
                hist_c[std::min<long>(cn, 64)]++;
                hist_l[std::min<long>(prevL, 64)]++;
            }
            const bool dis = prevR.valid() && prevR.beats(M);
            const bool viaR = best.key == prevR.key;
            if (dis != viaR) fprintf(stderr, "inconsistent at step %ld\n", steps);
            const bool bw = prevR.valid() && prevR.bw(s.n_bits) > M.bw(s.n_bits);
            const bool old = prevR.valid() && prevR.beats(prevV) && prevR.beats(Mnew);
            disjoint += dis;
            bwok += bw;
            oldvals += old;
            rec_dis.push_back(dis);
            rec_old.push_back(old);
            if (dis)
                ++run;
            else {
                if (run) ++runs, runsum += run;
                run = 0;
            }
        }
        const PairKey pick = best.key;
        Cand R, V;
        for (const auto &e : s.table) {
            const PairKey &k = e.first;
            const bool t0 = k.id0 == pick.id0 || k.id0 == pick.id1, t1 = k.id1 == pick.id0 || k.id1 == pick.id1;
            if (!t0 && !t1)
                R.offer(score_of(e), k);
            else if (t0 != t1)
                V.offer(score_of(e), k);
        }
        prevR = R;
        prevV = V;
        prevL = 0;
        if (R.valid())
            for (const auto &e : s.table) {
                const PairKey &k = e.first;
                const bool t0 = k.id0 == pick.id0 || k.id0 == pick.id1, t1 = k.id1 == pick.id0 || k.id1 == pick.id1;
                if (t0 == t1) continue;
                Cand c;
                c.offer(score_of(e), k);
                if (c.valid() && c.bw(s.n_bits) >= R.bw(s.n_bits)) ++prevL;
            }
        pA = pick.id0;
        pB = pick.id1;
        pN = (int64_t)s.expr.size();
        ++steps;
        substitute(s, pick, -1, -1);
        refresh_table(s, pick);
        if (steps % 500 == 0) fprintf(stderr, "step %ld table %zu disjoint %.3f oldvals %.3f\n", steps, s.table.size(), (double)disjoint / steps, (double)oldvals / steps);
    }
    const long T = (long)rec_dis.size();
    for (long i = 0; i < T; ++i) {
        const int q = (int)(4 * i / std::max<long>(T, 1));
        q_steps[q]++;
        q_dis[q] += rec_dis[i];
        q_old[q] += rec_old[i];
    }
    printf("n %d seed %d steps %ld classified %ld\n", n, seed, steps, T);
    printf("disjoint (exact fast path)      %.4f\n", (double)disjoint / T);
    printf("decided on the bound word       %.4f\n", (double)bwok / T);
    printf("certifiable from old values     %.4f\n", (double)oldvals / T);
    printf("mean run of fast steps          %.2f (%ld runs)\n", runs ? (double)runsum / runs : 0.0, runs);
    auto pct = [&](const std::vector<long> &h, const char *name) {
        long tot = 0, acc = 0;
        for (long v : h) tot += v;
        printf("%s (entries, not blocks: an upper bound): ", name);
        for (int i = 0; i <= 64; ++i) {
            acc += h[i];
            if (i == 0 || i == 1 || i == 2 || i == 4 || i == 8 || i == 16 || i == 32 || i == 63) printf("<=%d: %.4f  ", i, (double)acc / std::max<long>(tot, 1));
        }
        printf("\n");
    };
    pct(hist_c, "written/unchanged entries touching A,B,N with bound word >= R's");
    pct(hist_l, "old entries touching exactly one of A,B with bound word >= R's");
    for (int q = 0; q < 4; ++q)
        printf("quarter %d: disjoint %.4f oldvals %.4f\n", q, q_steps[q] ? (double)q_dis[q] / q_steps[q] : 0.0, q_steps[q] ? (double)q_old[q] / q_steps[q] : 0.0);
    return 0;
}
