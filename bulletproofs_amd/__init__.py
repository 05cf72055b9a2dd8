"""bulletproofs_amd: MI355X-native engine for the Bulletproofs verification hot path.

Host-side mirror of the reference crate's public surface for this path
(src/lib.rs:34-45 of dalek-cryptography/bulletproofs): RangeProof, LinearProof,
BulletproofGens, PedersenGens, ProofError, Transcript -- all of it a thin layer
over the C ABI of libbpgpu.so (include/bpgpu.h).  No CPU fallback exists.
"""
from ._lib import BpgpuError, Context, Pool, lib, LIB_PATH  # noqa: F401
from .api import (BulletproofGens, BulletproofGensShare, PedersenGens, RangeProof, LinearProof, Transcript, ProofError, VerificationError,  # noqa: F401,E402
                  FormatError, InvalidBitsize, InvalidGeneratorsLength)
