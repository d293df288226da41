// Host-only probe of the weight-gradient work split: how many distinct operand streams (a P or Q row stream = one node's rows of one
// buffer over the batch) every XCD's lanes touch.  With perfect L2 reuse inside an XCD each (stream, XCD) incidence is fetched once per
// window, so  sum_xcd distinct-stream bytes / unique-stream bytes  is the duplication factor the placement leaves to the fabric.
//   g++ -O1 -std=c++17 -shared -fPIC -o /tmp/gradw_sharing.so tools/probe/gradw_sharing.cpp
#include "../../morphsym_hgnn_amd/csrc/mshgnn_plan.hpp"
#include <map>
#include <set>
using namespace mshgnn;
extern "C" int probe(const mshgnn_desc* d, double* out, int32_t* lane_dump, int cap) {
    HostPlan hp;
    if (!compile_plan(d, hp)) { std::fprintf(stderr, "%s\n", hp.err.c_str()); return -1; }
    const int32_t* T = hp.tables.data();
    auto stream_bytes = [&](const int32_t* im, bool q) -> std::pair<std::array<int, 3>, double> {
        const int es = hp.esize;
        if (!q) return {{im[0], im[2], 0}, (double)H * es + (im[9] >= 0 ? 16 : 0)};
        if (im[4] >= 0) return {{im[3], im[5], 0}, (double)H * es};
        return {{im[3], im[5], im[6]}, (double)std::min(H, im[7]) * es};
    };
    std::map<std::array<int, 3>, double> uniq;
    std::vector<std::map<std::array<int, 3>, double>> per(8);
    const int n_pad = hp.n_lanes_pad;
    int k = 0;
    for (int b = 0; b < n_pad; ++b) {
        const int ln = T[hp.lane_order_off + b];
        if (ln < 0) continue;
        const int32_t* lh = T + hp.lane_off + ln * LANE_INTS;
        for (int it = lh[0]; it < lh[1]; ++it) {
            const int32_t* im = T + hp.item_off + it * ITEM_INTS;
            for (int q = 0; q < 2; ++q) { auto s = stream_bytes(im, q); s.first[0] += q ? 1000 : 0; uniq[s.first] = s.second; per[b % 8][s.first] = s.second; }
            if (lane_dump && k + 8 <= cap) { lane_dump[k++] = b % 8; lane_dump[k++] = lh[2]; lane_dump[k++] = im[0]; lane_dump[k++] = im[2]; lane_dump[k++] = im[3]; lane_dump[k++] = im[5]; lane_dump[k++] = im[6]; lane_dump[k++] = ln; }
        }
    }
    double u = 0, tot = 0; for (auto& kv : uniq) u += kv.second;
    for (auto& m : per) for (auto& kv : m) tot += kv.second;
    out[0] = u; out[1] = tot; out[2] = hp.n_lanes; out[3] = n_pad; out[4] = hp.n_parts; out[5] = hp.gw_ipl;
    for (int x = 0; x < 8; ++x) { double s = 0; for (auto& kv : per[x]) s += kv.second; out[6 + x] = s; }
    return k;
}
