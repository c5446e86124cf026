"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU checkers for the MI355X hot path:
  * liboracle.so         -- portable C restatement (np_oracle.c), built by `make -C oracle port`
  * _ref/libnp_ref.so    -- the reference's own hot path compiled in place from /root/reference
                            (`make -C oracle ref`; only possible where /root/reference exists)
  * _ref/libnp_ref_full.so -- the reference's own READ-LEVEL code on top of that (SquiggleRead::load_from_raw,
                            EventAlignmentRecord, calculate_methylation_for_read, create_modbam_record, align_read_to_ref),
                            `make -C oracle full`; binding in oracle/ref_full.py

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (nanopolish_amd/) never does; it fails loudly if its HIP library is missing.
"""
from .oracle_py import Oracle, RefOracle, load_models, have_ref  # noqa: F401
from . import workloads  # noqa: F401
