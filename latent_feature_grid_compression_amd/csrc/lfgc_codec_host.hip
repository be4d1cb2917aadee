// lfgc_codec_host.hip -- host-side part of the codebook construction (no device code).
//   lfgc_codec_ward_init_host   initial centres for the 1-D k-means of the checkpoint codec: agglomerative (Ward) merging
//                               of ADJACENT clusters of a sorted sample until k remain.  In one dimension this greedy
//                               merge is close to the optimal k-partition and, followed by the GPU Lloyd iterations on the
//                               full data, gives codebooks with a lower error than the reference's scikit-learn call
//                               (KMeans(n_init=4), model/model_utils.py:65-70) on the fixture tensors; it is
//                               deterministic, which the reference's unseeded clustering is not.
#include "../../include/lfgc.h"
#include <cstdint>
#include <queue>
#include <vector>

namespace {
struct Merge {
    double cost;
    int32_t left, right;
    uint32_t ver_left, ver_right;
    bool operator<(const Merge& o) const {              // min-heap on cost; ties: leftmost pair first (deterministic)
        return cost > o.cost || (cost == o.cost && left > o.left);
    }
};
}  // namespace

extern "C" int lfgc_codec_ward_init_host(const float* sorted_values, int64_t n, int k, float* centres) {
    if (!sorted_values || !centres) return LFGC_E_NULL;
    if (n < 1 || k < 1 || n > (int64_t)1 << 24) return LFGC_E_SHAPE;
    for (int64_t i = 1; i < n; ++i)
        if (sorted_values[i] < sorted_values[i - 1]) return LFGC_E_SHAPE;
    if (n <= k) {                                        // every value its own centre; pad by repeating the largest
        for (int j = 0; j < k; ++j) centres[j] = sorted_values[j < n ? j : n - 1];
        return LFGC_OK;
    }
    const int32_t m = (int32_t)n;
    std::vector<double> cnt(m, 1.0), sum(m);
    std::vector<int32_t> prev(m), next(m);
    std::vector<uint32_t> ver(m, 0);
    std::vector<char> alive(m, 1);
    for (int32_t i = 0; i < m; ++i) { sum[i] = sorted_values[i]; prev[i] = i - 1; next[i] = i + 1 < m ? i + 1 : -1; }
    auto cost = [&](int32_t a, int32_t b) {
        const double d = sum[a] / cnt[a] - sum[b] / cnt[b];
        return cnt[a] * cnt[b] / (cnt[a] + cnt[b]) * d * d;
    };
    std::priority_queue<Merge> heap;
    for (int32_t i = 0; i + 1 < m; ++i) heap.push(Merge{cost(i, i + 1), i, i + 1, 0u, 0u});
    int32_t clusters = m;
    while (clusters > k && !heap.empty()) {
        const Merge t = heap.top();
        heap.pop();
        const int32_t a = t.left, b = t.right;
        if (!alive[a] || !alive[b] || ver[a] != t.ver_left || ver[b] != t.ver_right || next[a] != b) continue;   // stale
        cnt[a] += cnt[b]; sum[a] += sum[b]; alive[b] = 0; ++ver[a];
        const int32_t nb = next[b];
        next[a] = nb;
        if (nb >= 0) { prev[nb] = a; heap.push(Merge{cost(a, nb), a, nb, ver[a], ver[nb]}); }
        const int32_t pa = prev[a];
        if (pa >= 0) heap.push(Merge{cost(pa, a), pa, a, ver[pa], ver[a]});
        --clusters;
    }
    int j = 0;
    for (int32_t i = 0; i < m && j < k; ++i)
        if (alive[i]) centres[j++] = (float)(sum[i] / cnt[i]);
    for (; j < k; ++j) centres[j] = centres[j - 1];
    return LFGC_OK;
}
