#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/pmc_configs.sh: the HBM traffic per launch of each config's dominant kernel
(2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md 'HBM'), under the key bench.py looks up -- "<pmf_kernel_stats name>@<m>x<n>x<k>/<ranks>".

    python tools/traffic_json.py gpurun_out/r05_traffic.json r05
"""
import json
import sys

src, tag = sys.argv[1], sys.argv[2]
raw = json.load(open(src))
# (config, kernel name as rocprofv3 prints it up to the argument list) -> (name as pmf_kernel_stats reports it, shape)
MAP = {
    ("cfg4", "void k_nmf_fused<4, 4, 0, 1>"): ("k_nmf_fused<4,4>", "1048576x256x64"),
    ("cfg2", "void k_nmf_fused<2, 4, 0, 2>"): ("k_nmf_fused<2,4,SPLIT 2>", "65536x512x32"),
    ("cfg3", "void k_nnqp_quad<16, 12, false>"): ("k_nnqp_quad(update_w)", "262144x1024x64"),
    ("cfg5", "void k_csr_w_blocks<8>"): ("k_csr_w_blocks(W = V M)", "4194304x128x128"),
    # cfg3's two long products (not the line's `roofline` kernel: recorded for DESIGN 3.3 / the verdict's traffic ratios)
    ("cfg3", "void k_rowgemm_stream<4, 4, 0, false>"): ("k_rowgemm_stream<4,4,store>", "262144x1024x64"),
    ("cfg3", "void k_colgemm_stream<4, true>"): ("k_colgemm_stream<4,true>", "262144x1024x64"),
}
out = {"_provenance": "round %s (tools/pmc_configs.sh %s -> profiles/%s_pmc_summary.csv): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                      "separate passes over bench.py per config; bytes = 2 * FETCH_SIZE (gfx950: FETCH_SIZE reports half of a wide "
                      "coalesced streaming read, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE, mean per dispatch of the config's dominant "
                      "kernel; keys = <pmf_kernel_stats name>@<m>x<n>x<k>/<ranks> (tools/traffic_json.py)" % (tag, tag, tag),
       "_raw": raw}
for (cfg, kern), (name, shape) in MAP.items():
    v = raw.get("%s|%s" % (cfg, kern))
    if v is None:
        print("missing:", cfg, kern)
        continue
    out["%s@%s/1" % (name, shape)] = v["hbm_bytes_per_launch"]
    print("%-45s %14.0f B/launch" % ("%s@%s/1" % (name, shape), v["hbm_bytes_per_launch"]))
json.dump(out, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
