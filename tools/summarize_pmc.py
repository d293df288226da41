"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) into per-kernel HBM-side bytes per launch.
gfx950 corrections (MI355X_MICROARCH.md, section HBM): counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of wide coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-byte streaming stores."""
import collections, csv, glob, json, sys

def per_kernel(pattern, counter):
    agg = collections.defaultdict(list)
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v), len(v)) for k, v in agg.items()}      # (total, launches) per kernel name

def short(name):
    for key, s in [("k_prep", "prep"), ("k_enc_fwd", "enc_fwd"), ("k_enc_x3", "enc_fwd"), ("k_gstep", "gstep"), ("k_ggradw", "ggradw"), ("k_stack_fwd", "stack_fwd"), ("k_stack_bwd", "stack_bwd"), ("k_slab_step", "stack_step"), ("k_stack_step", "stack_step"), ("k_slab_fwd", "stack_fwd"), ("k_slab_bwd", "stack_bwd"), ("k_eng_fwd", "stack_fwd"), ("k_eng_bwd", "stack_bwd"), ("k_wide_fwd", "stack_fwd"), ("k_wide_bwd", "stack_bwd"),
                   ("k_layer_fwd", "layer_fwd"), ("k_dec_fwd", "dec_fwd"),
                   ("k_dec_bwd", "dec_bwd"), ("k_layer_bwd", "layer_bwd"), ("k_gradw", "gradw"), ("k_finalize", "finalize"), ("k_mse", "mse")]:
        if key in name:
            return s
    return None

if __name__ == "__main__":
    fetch = per_kernel(sys.argv[1] + "/**/*counter_collection.csv", "FETCH_SIZE")
    write = per_kernel(sys.argv[2] + "/**/*counter_collection.csv", "WRITE_SIZE")
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])      # short name -> [fetch KiB, write KiB, launches]: the instantiations of one kernel family are pooled
    for k, (v, n) in fetch.items():
        s = short(k)
        if s:
            tot[s][0] += v; tot[s][2] += n
            tot[s][1] += write.get(k, (0.0, 0))[0]
    out = {}
    for s, (f, w, n) in tot.items():
        out[s] = {"fetch_bytes": 2.0 * f / n * 1024.0, "write_bytes": w / n * 1024.0, "launches": n}
        out[s]["hbm_bytes"] = out[s]["fetch_bytes"] + out[s]["write_bytes"]
    import os
    cfile = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")      # written before the snapshot leaves the build container
    commit = open(cfile).read().strip() if os.path.exists(cfile) else None
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_hash
    json.dump({**({"commit": commit} if commit else {}), "source_hash": source_hash(), "note": "per launch, averaged over launches (layer_fwd / layer_bwd: averaged over the L layers); "
                       "FETCH_SIZE x2 (gfx950), KiB -> bytes", "kernels": out}, sys.stdout, indent=1)
