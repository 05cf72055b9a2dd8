"""bulletproofs_amd: MI355X-native engine for the Bulletproofs verification hot path.

Host-side mirror of the reference crate's public surface for this path
(src/lib.rs:34-45 of dalek-cryptography/bulletproofs): RangeProof, LinearProof,
BulletproofGens, PedersenGens, ProofError, Transcript -- all of it a thin layer
over the C ABI of libbpgpu.so (include/bpgpu.h).  No CPU fallback exists.
"""
import os as _os

# The loader's duty (include/bpgpu.h, "pool"): the ROCm runtime reads GPU_MAX_HW_QUEUES at the process's first HIP call, and the
# pool's lanes need 8..16 hardware queues.  Import this package before anything initialises HIP (torch.cuda, another HIP library);
# bpgpu_pool_create probes the device and fails loudly when it was too late.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._lib import BpgpuError, Context, Pool, lib, LIB_PATH  # noqa: F401
from .api import (BulletproofGens, BulletproofGensShare, PedersenGens, RangeProof, LinearProof, Transcript, ProofError, VerificationError,  # noqa: F401,E402
                  FormatError, InvalidBitsize, InvalidGeneratorsLength)
