#!/usr/bin/env python3
"""The RNA pore model the device path needs as a fixture: r9.4_70bps / u_to_t_rna / template / 5-mer as the reference's
PoreModelSet holds it (src/nanopolish_squiggle_read.cpp:206-213 picks it for a direct-RNA read), exported from the reference compiled in
place (oracle/_ref/libnp_ref.so).  Run in the build container:  python tests/gen_golden_rna.py  -> tests/golden/models_r9.4_70bps_rna.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import RefOracle  # noqa: E402


def main():
    ref = RefOracle()
    m = ref.model("u_to_t_rna", k=5, kit=b"r9.4_70bps")
    assert len(m["level_mean"]) == 1024
    np.savez_compressed(os.path.join(HERE, "golden", "models_r9.4_70bps_rna.npz"),
                        **{"u_to_t_rna_" + f: m[f] for f in ("level_mean", "level_stdv", "level_log_stdv")})
    print("u_to_t_rna 5-mer: %d states, mean level %.2f pA" % (len(m["level_mean"]), float(np.mean(m["level_mean"]))))


if __name__ == "__main__":
    main()
