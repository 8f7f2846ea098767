// Model of level-1 walk lengths with and without a bound inherited from an earlier position (CPU, exact five-byte
// chains; no hash collisions, no chain changes during a walk).  For every position p of the input: n5[p] = members
// of the chain of data[p..p+5) inside the 32 KiB window, L[p] = longest match.  A position that knows a candidate of
// b = L[p-k] - k >= 6 bytes (found by position p - k) may walk the chain of any offset o <= b - 5 instead of its
// own; the model takes the shortest of them (o <= omax).  "phases": the order positions would be searched in so
// that the earlier position is finished first -- 1: every position inherits from p-1 (a serial order), 2: odd
// positions from p-1, 4: also p = 2 mod 4 from p-2, 8: also p = 4 mod 8 from p-4.
// usage: g++ -O2 -o sim_inherit tools/sim_inherit.cpp && ./sim_inherit FILE
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unordered_map>
#include <vector>
int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> d;
    {
        uint8_t buf[65536];
        size_t r;
        while ((r = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + r);
    }
    fclose(f);
    const int n = (int)d.size(), W = 32768;
    std::vector<int> prev(n, -1), n5(n, 0), L(n, 0);
    {
        std::unordered_map<uint64_t, int> last;
        for (int p = 0; p + 5 <= n; p++) {
            uint64_t k = 0;
            memcpy(&k, &d[p], 5);
            auto it = last.find(k);
            if (it != last.end()) prev[p] = it->second;
            last[k] = p;
        }
    }
    for (int p = 0; p + 5 <= n; p++) {
        int c = prev[p], cnt = 0, best = 0;
        while (c >= 0 && p - c <= W) {
            cnt++;
            int m = 0;
            const int cap = std::min(257, n - p);
            while (m < cap && d[c + m] == d[p + m]) m++;
            best = std::max(best, m);
            c = prev[c];
        }
        n5[p] = cnt;
        L[p] = best;
    }
    long npos = 0;
    double base = 0;
    for (int p = W; p + 300 < n; p++) npos++, base += n5[p];
    printf("positions %ld, chain members per search %.2f\n", npos, base / npos);
    for (int omax : {3, 7, 255})
        for (int ph : {1, 2, 4, 8}) {
            double tot = 0;
            for (int p = W; p + 300 < n; p++) {
                int k = 1;
                if (ph > 1) k = (p & 1) ? 1 : (ph > 2 && (p & 2)) ? 2 : (ph > 4 && (p & 4)) ? 4 : 0;
                int cost = n5[p];
                if (k) {
                    const int b = L[p - k] - k;
                    if (b >= 6)
                        for (int o = 1; o <= b - 5 && o <= omax; o++) cost = std::min(cost, n5[p + o]);
                }
                tot += cost;
            }
            printf("offsets up to %3d, phases %d: %.2f\n", omax, ph, tot / npos);
        }
}
