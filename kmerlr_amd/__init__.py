"""kmerlr_amd -- MI355X (gfx950) implementation of kmerLr's L1 logistic-regression hot path.

The product is libkmerlr_hip.so (C ABI in include/kmerlr.h); this package is the thin Python
plumbing around it (ctypes binding, torch.distributed launcher glue).  There is no CPU fallback.
"""
from ._capi import (Context, Matrix, KmerClasses, KlrError, gradient, gradient_dense, loss, log_pdf, select,  # noqa: F401
                    select_fixed, sweep_select, SweepStep, transform_fit, cd_solve, cd_sweep_rate, synth_csr, rccl_unique_id, coeff_sub2ind, lib, LIB_PATH, SIGNATURES, version, kernel_hash)
from ._driver import Estimator, hip_estimator, Config, Backend, bind_driver  # noqa: F401,E402
