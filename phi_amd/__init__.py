"""phi_amd -- MI355X-native hot path of at-cg/PHI (haplotype inference from an acyclic pangenome GFA).

The package is a thin host layer over two native libraries built in-tree:
  libphi_amd.so   hand-written HIP kernels for gfx950 behind the C ABI of include/phi_amd.h
  libphi_host.so  host-side GFA / FASTQ readers (include/phi_host.h)
There is no CPU fallback: importing works anywhere, using the hot path needs the library and a GPU.
"""
from ._capi import (PHI_ERR_DEVICE, PHI_ERR_INVALID, PHI_ERR_NOMEM, PHI_ERR_OVERFLOW, PHI_ERR_STATE,  # noqa: F401
                    PHI_ERR_UNSUPPORTED, PHI_ERR_WALK, PHI_FLAG_MIXED, PHI_FLAG_QCLP, PHI_OK)
from .context import Context, PhiError  # noqa: F401

__all__ = ["Context", "PhiError"]
