"""Host-side model of k_ggradw's operand traffic on the generic engine (configs[4]): reads the plan's own tables (libmshgnn_hostplan.so), lists targets / super-units /
steps and replays the launch as a list schedule over 8 XCD queues to estimate what a per-XCD L2 of a given size can share.  CPU only.
usage: python tools/ggradw_traffic_model.py [--config synth32 --layers 6 --hidden 512 --batch 1024]"""
import argparse, ctypes as C, os, sys
from collections import defaultdict, OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from morphsym_hgnn_amd import engine as eng

SUNIT_INTS, GITEM_INTS, SRC_INTS = 12, 8, 4
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="synth32"); ap.add_argument("--layers", type=int, default=6); ap.add_argument("--hidden", type=int, default=512)
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--kw", type=int, default=64); ap.add_argument("--l2mb", type=float, default=4.0)
args = ap.parse_args()
spec = bench.build_spec(args.layers, args.config, args.hidden)
lib = C.CDLL(os.path.join(ROOT, "morphsym_hgnn_amd", "libmshgnn_hostplan.so"))
lib.mshgnn_hostplan_gen_tables.argtypes = [C.POINTER(eng.MshgnnDesc), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32)]
h = eng._DescHolder(spec, eng.DTYPE_CODES["bf16"])
meta = (C.c_int32 * 16)()
n = lib.mshgnn_hostplan_gen_tables(C.byref(h.desc), None, 0, meta)
assert n > 0
T = (C.c_int32 * n)(); lib.mshgnn_hostplan_gen_tables(C.byref(h.desc), T, n, meta)
T = list(T)
sunit_off, su_order_off, item_off, src_off, n_sunits, n_parts, su_os, n_units = meta[:8]
Hd, B = args.hidden, args.batch
print(f"n_sunits {n_sunits}  n_parts {n_parts}  su_os {su_os}  n_units {n_units}  workgroups {n_sunits * n_parts}  slabs {n_units * n_parts * (128 * 128 + 128) * 4 / 1e6:.1f} MB")
su = [T[sunit_off + i * SUNIT_INTS: sunit_off + (i + 1) * SUNIT_INTS] for i in range(n_sunits)]
order = T[su_order_off: su_order_off + n_sunits]
def item(i): return T[item_off + i * GITEM_INTS: item_off + (i + 1) * GITEM_INTS]
def src(i): return T[src_off + i * SRC_INTS: src_off + (i + 1) * SRC_INTS]
# streams of a super-unit step: (buf, node, col0) x 256 columns x rows
def su_streams(s):
    out = []
    for it in range(s[4], s[5]):
        im = item(it)
        ps = [("P", im[0], im[1], s[6])]
        if im[3] == 1:
            qs = [("R", src(im[4])[0], src(im[4])[1], s[7])]
        else:
            qs = [("Q", src(im[4] + k)[0], src(im[4] + k)[1], s[7]) for k in range(im[5])]
        out.append((ps, qs))
    return out
steps = [s[5] - s[4] for s in su]
lean = [s[9] & 1 for s in su]
print("items per super-unit:", sorted(set(steps)), " lean", sum(lean), "general", n_sunits - sum(lean))
rowb = 256 * 2   # bytes of a 256-column piece of a bf16 row
req = 0; distinct = set()
for s in su:
    for ps, qs in su_streams(s):
        for st in ps + qs:
            w = min(256, s[8]) if st[0] != "P" else 256
            req += w * 2 * B; distinct.add((st, w))
dist = sum(w * 2 * B for _, w in distinct)
print(f"requested {req / 1e9:.2f} GB   distinct (buf,node,256-col piece) {dist / 1e9:.2f} GB   ratio {req / dist:.2f}")
# replay: block b -> (sunit order[b % n_sunits], part b // n_sunits) on XCD b % 8, CU slots per XCD = 32, one workgroup per CU; every step takes 1 (lean) or 2 (general) time units;
# an XCD-level LRU of l2mb holds (stream piece, window chunk) lines of kw windows
nchunks = (B + args.kw - 1) // args.kw
blocks = [(order[b % n_sunits], b // n_sunits) for b in range(n_sunits * n_parts)]
import heapq
for l2mb in (args.l2mb, 32.0):      # 32 MB per XCD: what a 256 MB Infinity Cache shared by 8 XCDs amounts to
    l2cap = int(l2mb * 1e6)
    hbm = 0; hits = 0
    queues = [[] for _ in range(8)]
    for b, blk in enumerate(blocks): queues[b % 8].append(blk)
    for x in range(8):
        lru = OrderedDict(); used = 0
        # event simulation: 32 CUs, each runs one workgroup at a time; a workgroup's steps are (item, chunk) pairs, item-major over its part's chunks
        q = list(queues[x]); qi = 0
        cus = []      # (time, cu, iterator state)
        def wg_steps(si, part):
            s = su[si]; c0 = part * nchunks // n_parts; c1 = (part + 1) * nchunks // n_parts
            for (ps, qs) in su_streams(s):
                for c in range(c0, c1):
                    yield [(st, c, (256 if st[0] == "P" else min(256, s[8])) * 2 * args.kw) for st in ps + qs], (1 if s[9] & 1 else 2)
        heap = []
        for cu in range(32):
            if qi < len(q): it = wg_steps(*q[qi]); qi += 1; heapq.heappush(heap, (0.0, cu, id(it), it))
        while heap:
            t, cu, _, it = heapq.heappop(heap)
            try:
                lines, dur = next(it)
            except StopIteration:
                if qi < len(q): it = wg_steps(*q[qi]); qi += 1; heapq.heappush(heap, (t, cu, id(it), it))
                continue
            for (st, c, nb) in lines:
                key = (st, c)
                if key in lru: lru.move_to_end(key); hits += nb
                else:
                    hbm += nb; lru[key] = nb; used += nb
                    while used > l2cap: _, ev = lru.popitem(last=False); used -= ev
            heapq.heappush(heap, (t + dur, cu, id(it), it))
    print(f"per-XCD LRU of {l2mb:.0f} MB: beyond-cache bytes {hbm / 1e9:.2f} GB, hit rate {hits / (hits + hbm):.2f}")
# what the general super-units are
import struct
kinds = defaultdict(int)
for s in su:
    if s[9] & 1: continue
    im = item(s[4]); sr = src(im[4])
    kinds[("raw" if im[3] == 1 else f"act nsrc={im[5]} scale={struct.unpack('f', struct.pack('i', sr[3]))[0]:.3f} mask={sr[2]}", "qn", s[8], "items", s[5] - s[4])] += 1
print("general super-units:", dict(kinds))
