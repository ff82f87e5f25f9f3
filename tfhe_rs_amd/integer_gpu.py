"""Host-side mirror of tfhe/src/integer/gpu for the radix operations the backend wires ("next" row N1):
CudaServerKey (keys + parameters on the device), CudaUnsignedRadixCiphertext (here: a BATCH of
integers, [integer][block] on the device) and the operations

    unchecked_add_assign / add_assign   integer/gpu/server_key/radix/add.rs
    propagate_single_carry_assign       integer/gpu/mod.rs (cuda_backend_propagate_single_carry_assign)
    mul_assign                          integer/gpu/server_key/radix/mul.rs
    apply_lookup_table                  integer/gpu/mod.rs (cuda_backend_apply_univariate_lut)

Each call is scratch -> launch -> cleanup through the C ABI, as the Rust wrappers do
(integer/gpu/mod.rs).  No CPU fallback: everything runs in libtfhe_hip_backend.so.
"""
import ctypes as C

import numpy as np

from . import ffi
from .core_crypto_gpu import CudaLweBootstrapKey, CudaLweKeyswitchKey, CudaVec, _lib

U64 = np.uint64
PBS_TYPE_MULTI_BIT, PBS_TYPE_CLASSICAL = 0, 1          # pbs/pbs_enums.h:4
OUTPUT_FLAG_NONE, OUTPUT_FLAG_OVERFLOW, OUTPUT_FLAG_CARRY = 0, 1, 2   # integer/integer.h:39


class CudaServerKey:
    """integer/gpu/server_key/mod.rs:26-60: keyswitch key, bootstrap key and the shortint parameters."""

    def __init__(self, ksk: CudaLweKeyswitchKey, bsk: CudaLweBootstrapKey, message_modulus, carry_modulus):
        assert ksk.output_key_lwe_dimension == bsk.input_lwe_dimension
        assert ksk.input_key_lwe_dimension == bsk.output_lwe_dimension
        self.key_switching_key, self.bootstrapping_key = ksk, bsk
        self.message_modulus, self.carry_modulus = int(message_modulus), int(carry_modulus)

    # ---- FFI views
    def _bsk_params(self):
        b = self.bootstrapping_key
        g = getattr(b, "grouping_factor", 0)   # a CudaLweMultiBitBootstrapKey selects the multi-bit PBS
        return ffi.CudaLweBootstrapKeyParamsFFI(b.input_lwe_dimension, b.glwe_dimension, b.polynomial_size,
                                                b.decomp_base_log, b.decomp_level_count, b.output_lwe_dimension,
                                                PBS_TYPE_MULTI_BIT if g else PBS_TYPE_CLASSICAL, g)

    def _ksk_params(self):
        k = self.key_switching_key
        return ffi.CudaLweKeyswitchKeyParamsFFI(k.input_key_lwe_dimension, k.output_key_lwe_dimension,
                                                k.decomp_base_log, k.decomp_level_count)

    def _key_ptrs(self, streams=None):
        """One key replica per stream of the set (gpu/ffi.rs passes `ksks` / `bsks` arrays of per-GPU pointers:
        integer/gpu/mod.rs); the keys must have been converted with a stream set at least as large."""
        n = len(streams) if streams is not None else 1
        kv, bv = self.key_switching_key.d_vecs, self.bootstrapping_key.d_vecs
        assert len(kv) >= n and len(bv) >= n, "server key has fewer GPU replicas than the stream set has streams"
        ksks = (C.c_void_p * n)(*[v.ptr for v in kv[:n]])
        bsks = (C.c_void_p * n)(*[v.ptr for v in bv[:n]])
        return ksks, bsks

    @staticmethod
    def _streams(streams):
        ptrs = (C.c_void_p * len(streams))(*streams.ptr)
        idx = (C.c_uint32 * len(streams))(*streams.gpu_indexes)
        return ffi.CudaStreamsFFI(ptrs, idx, len(streams)), (ptrs, idx)

    def _noise_reduction(self):
        return 1 if getattr(self.bootstrapping_key, "ms_noise_reduction", False) else 0

    # ---- operations
    def apply_lookup_table(self, ct, lut, streams, degree=None):
        """Every block of every integer goes through KS -> PBS with `lut` (a (k+1)*N accumulator)."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        lut = np.ascontiguousarray(lut, dtype=U64)
        mem = C.c_void_p()
        n = ct.total_blocks
        _lib().scratch_cuda_apply_univariate_lut_64_async(
            s, C.byref(mem), lut.ctypes.data_as(C.c_void_p), self._bsk_params(), self._ksk_params(), n,
            self.message_modulus, self.carry_modulus, degree if degree is not None else self.message_modulus - 1, True,
            self._noise_reduction())
        out = CudaUnsignedRadixCiphertext.zeros_like(ct, streams)
        _lib().cuda_apply_univariate_lut_64_async(s, C.byref(out._ffi()), C.byref(ct._ffi()), mem, ksks, bsks)
        _lib().cleanup_cuda_apply_univariate_lut_64(s, C.byref(mem))
        return out

    def apply_many_lookup_table(self, ct, many_lut, num_luts, lut_stride, streams, degree=None):
        """integer/gpu/mod.rs cuda_backend_apply_many_univariate_lut: ONE keyswitch and ONE bootstrap per block evaluate the
        `num_luts` functions packed in `many_lut` (shortint ManyLookupTable: sub-tables of `lut_stride` coefficients,
        shortint/engine/mod.rs:169-254).  Returns a ciphertext of num_luts * blocks blocks: function t of block s of
        the input at flat block t * total_blocks + s."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        many_lut = np.ascontiguousarray(many_lut, dtype=U64)
        mem = C.c_void_p()
        n = ct.total_blocks
        _lib().scratch_cuda_apply_many_univariate_lut_64_async(
            s, C.byref(mem), many_lut.ctypes.data_as(C.c_void_p), self._bsk_params(), self._ksk_params(), n,
            self.message_modulus, self.carry_modulus, num_luts, degree if degree is not None else self.message_modulus - 1,
            True, self._noise_reduction())
        out = CudaUnsignedRadixCiphertext(CudaVec(num_luts * n * (ct.lwe_dimension + 1), streams), num_luts * ct.num_integers,
                                          ct.num_blocks, ct.lwe_dimension)
        _lib().cuda_apply_many_univariate_lut_64_async(s, C.byref(out._ffi()), C.byref(ct._ffi()), mem, ksks, bsks,
                                                       num_luts, lut_stride)
        _lib().cleanup_cuda_apply_many_univariate_lut_64(s, C.byref(mem))
        return out

    def unchecked_add_assign(self, lhs, rhs, streams):
        """Block-wise LWE addition, no carry handling (radix/add.rs unchecked_add_assign)."""
        assert lhs.total_blocks == rhs.total_blocks
        _lib().cuda_add_lwe_ciphertext_vector_inplace_64(streams.ptr[0], streams.gpu_indexes[0], C.byref(lhs._ffi()),
                                                         C.byref(rhs._ffi()))

    def _carry_blocks(self, ct, carry, streams):
        """The reference's wrappers ALWAYS hand carry_in / carry_out radix structs to the backend, whether
        or not uses_carry / requested_flag select them (integer/gpu/ffi.rs:2194-2237): one block per
        integer, zero (trivial) unless the caller supplies one."""
        if carry is not None:
            assert carry.total_blocks == ct.num_integers and carry.lwe_dimension == ct.lwe_dimension
            return carry
        return CudaUnsignedRadixCiphertext(CudaVec(ct.num_integers * (ct.lwe_dimension + 1), streams),
                                           ct.num_integers, 1, ct.lwe_dimension)

    def propagate_single_carry_assign(self, ct, streams, carry_in=None, want_carry_out=False):
        """integer/gpu/mod.rs propagate_single_carry_assign: carry_in (one block per integer, 0/1) enters
        block 0 when given; with want_carry_out (OutputFlag::Carry) the carry leaving the last block is
        returned as a one-block-per-integer ciphertext."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        flag = OUTPUT_FLAG_CARRY if want_carry_out else OUTPUT_FLAG_NONE
        cin, cout = self._carry_blocks(ct, carry_in, streams), self._carry_blocks(ct, None, streams)
        _lib().hip_integer_scratch_batch(ct.num_integers)
        _lib().scratch_cuda_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), ct.num_blocks, self.message_modulus,
            self.carry_modulus, flag, True, self._noise_reduction())
        _lib().cuda_propagate_single_carry_64_inplace_async(s, C.byref(ct._ffi()), C.byref(cout._ffi()),
                                                            C.byref(cin._ffi()), mem, bsks, ksks, flag,
                                                            1 if carry_in is not None else 0)
        _lib().cleanup_cuda_propagate_single_carry_64_inplace(s, C.byref(mem))
        return cout if want_carry_out else None

    def add_assign(self, lhs, rhs, streams, carry_in=None, want_carry_out=False, want_overflow=False):
        """lhs += rhs (+ carry_in) on clean (carry-free) operands: block additions, then one carry propagation.
        want_carry_out (OutputFlag::Carry): returns the carry leaving the last block; want_overflow
        (OutputFlag::Overflow, signed integers): returns the signed-overflow flag of the addition instead
        (integer/gpu/server_key/radix/add.rs signed_overflowing_add)."""
        assert not (want_carry_out and want_overflow)
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        flag = OUTPUT_FLAG_OVERFLOW if want_overflow else OUTPUT_FLAG_CARRY if want_carry_out else OUTPUT_FLAG_NONE
        cin, cout = self._carry_blocks(lhs, carry_in, streams), self._carry_blocks(lhs, None, streams)
        _lib().hip_integer_scratch_batch(lhs.num_integers)
        _lib().scratch_cuda_add_and_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), lhs.num_blocks, self.message_modulus,
            self.carry_modulus, flag, True, self._noise_reduction())
        _lib().cuda_add_and_propagate_single_carry_64_inplace_async(s, C.byref(lhs._ffi()), C.byref(rhs._ffi()),
                                                                    C.byref(cout._ffi()), C.byref(cin._ffi()), mem,
                                                                    bsks, ksks, flag, 1 if carry_in is not None else 0)
        _lib().cleanup_cuda_add_and_propagate_single_carry_64_inplace(s, C.byref(mem))
        return cout if (want_carry_out or want_overflow) else None

    def sub_assign(self, lhs, rhs, streams, want_carry_out=False):
        """lhs -= rhs (mod 2^bits) on clean operands: rhs negated with its correcting term, block additions, one carry
        propagation (integer/gpu/server_key/radix/sub.rs:347-400).  want_carry_out: the carry leaving the last block of
        lhs + (2^bits - rhs), i.e. 1 when NO borrow occurred."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        flag = OUTPUT_FLAG_CARRY if want_carry_out else OUTPUT_FLAG_NONE
        cin, cout = self._carry_blocks(lhs, None, streams), self._carry_blocks(lhs, None, streams)
        _lib().hip_integer_scratch_batch(lhs.num_integers)
        _lib().scratch_cuda_sub_and_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), lhs.num_blocks, self.message_modulus,
            self.carry_modulus, flag, True, self._noise_reduction())
        _lib().cuda_sub_and_propagate_single_carry_64_inplace_async(s, C.byref(lhs._ffi()), C.byref(rhs._ffi()),
                                                                    C.byref(cout._ffi()), C.byref(cin._ffi()), mem,
                                                                    bsks, ksks, flag, 0)
        _lib().cleanup_cuda_sub_and_propagate_single_carry_64_inplace(s, C.byref(mem))
        return cout if want_carry_out else None

    def unsigned_overflowing_sub_assign(self, lhs, rhs, streams):
        """lhs -= rhs; returns the borrow, one boolean block per integer (radix/sub.rs unsigned_overflowing_sub)."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        cin, cout = self._carry_blocks(lhs, None, streams), self._carry_blocks(lhs, None, streams)
        _lib().hip_integer_scratch_batch(lhs.num_integers)
        _lib().scratch_cuda_integer_overflowing_sub_64_inplace_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), lhs.num_blocks, self.message_modulus,
            self.carry_modulus, 1, True, self._noise_reduction())
        _lib().cuda_integer_overflowing_sub_64_inplace_async(s, C.byref(lhs._ffi()), C.byref(rhs._ffi()), C.byref(cout._ffi()),
                                                             C.byref(cin._ffi()), mem, bsks, ksks, 1, 0)
        _lib().cleanup_cuda_integer_overflowing_sub_64_inplace(s, C.byref(mem))
        return cout

    def unchecked_neg(self, ct, streams):
        """-ct of ONE integer, levelled (radix/neg.rs unchecked_neg): blocks z - b with the borrowed unit handed on; the
        result carries degrees above the message modulus (propagate before the next bootstrap-free operation)."""
        assert ct.num_integers == 1
        s, keep = self._streams(streams)
        out = CudaUnsignedRadixCiphertext.zeros_like(ct, streams)
        _lib().cuda_negate_ciphertext_64(s, C.byref(out._ffi()), C.byref(ct._ffi()), self.message_modulus,
                                         self.carry_modulus, ct.total_blocks)
        return out

    def unchecked_scalar_add_assign(self, ct, clear_blocks, streams):
        """ct[i] += clear_blocks[i] (plaintext addition on the bodies; radix/scalar_add.rs unchecked_scalar_add_assign)."""
        import numpy as np
        s, keep = self._streams(streams)
        h = np.ascontiguousarray(np.asarray(clear_blocks, dtype=np.uint64))
        d = CudaVec(max(1, h.size), streams)
        d.copy_from_cpu_async(h, streams)
        _lib().cuda_scalar_addition_ciphertext_64_inplace(s, C.byref(ct._ffi()), d.ptr, h.ctypes.data_as(C.c_void_p),
                                                          h.size, self.message_modulus, self.carry_modulus)
        streams.synchronize()

    def bitnot_assign(self, ct, streams):
        """Bitwise NOT of clean blocks, levelled (radix/bitwise_op.rs unchecked_bitnot_assign)."""
        s, keep = self._streams(streams)
        _lib().cuda_bitnot_ciphertext_64(s, C.byref(ct._ffi()), self.message_modulus, self.message_modulus,
                                         self.carry_modulus)

    def bitop_assign(self, lhs, rhs, op, streams):
        """lhs <- lhs (and | or | xor) rhs, one bivariate bootstrap per block pair (radix/bitwise_op.rs
        unchecked_bitop_assign); op in {"and", "or", "xor"}."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        code = {"and": 0, "or": 1, "xor": 2}[op]
        _lib().hip_integer_scratch_batch(1)
        _lib().scratch_cuda_integer_bitop_inplace_64_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), lhs.total_blocks, self.message_modulus,
            self.carry_modulus, code, True, self._noise_reduction())
        _lib().cuda_integer_bitop_inplace_64_async(s, C.byref(lhs._ffi()), C.byref(rhs._ffi()), mem, bsks, ksks)
        _lib().cleanup_cuda_integer_bitop_inplace_64(s, C.byref(mem))

    def scalar_bitop_assign(self, ct, clear_blocks, op, streams):
        """ct <- ct (and | or | xor) scalar, the scalar given as its clear blocks, least significant first
        (radix/scalar_bitwise_op.rs): a univariate table per clear value; AND clears the blocks past the scalar's."""
        import numpy as np
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        code = {"and": 3, "or": 4, "xor": 5}[op]
        h = np.ascontiguousarray(np.asarray(clear_blocks, dtype=np.uint64))
        d = CudaVec(max(1, h.size), streams)
        d.copy_from_cpu_async(h, streams)
        _lib().hip_integer_scratch_batch(1)
        _lib().scratch_cuda_integer_scalar_bitop_inplace_64_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), ct.total_blocks, self.message_modulus,
            self.carry_modulus, code, True, self._noise_reduction())
        _lib().cuda_integer_scalar_bitop_inplace_64_async(s, C.byref(ct._ffi()), d.ptr, h.ctypes.data_as(C.c_void_p),
                                                          h.size, mem, bsks, ksks)
        _lib().cleanup_cuda_integer_scalar_bitop_inplace_64(s, C.byref(mem))
        streams.synchronize()

    def full_propagate_assign(self, ct, streams):
        """Block after block: message kept, carry added to the next block (radix/mod.rs full_propagate_parallelized's
        sequential form, integer.cuh:1924-1983); ONE integer; the last block's carry is dropped."""
        assert ct.num_integers == 1
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        _lib().scratch_cuda_full_propagation_64_inplace_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), self.message_modulus, self.carry_modulus, True,
            self._noise_reduction())
        _lib().cuda_full_propagation_64_inplace_async(s, C.byref(ct._ffi()), mem, ksks, bsks, ct.total_blocks)
        _lib().cleanup_cuda_full_propagation_64_inplace(s, C.byref(mem))

    COMPARISONS = {"eq": 0, "ne": 1, "gt": 2, "ge": 3, "lt": 4, "le": 5, "max": 6, "min": 7}  # integer.h:24-33

    def compare(self, lhs, rhs, op, streams):
        """Unsigned comparison of ONE integer pair (radix/comparison.rs unchecked_{eq,ne,gt,ge,lt,le,max,min}): a boolean
        block for eq ... le, an integer for max / min."""
        assert lhs.num_integers == rhs.num_integers == 1 and lhs.num_blocks == rhs.num_blocks
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        code = self.COMPARISONS[op]
        L, w = lhs.num_blocks, lhs.lwe_dimension + 1
        out = (CudaUnsignedRadixCiphertext.zeros_like(lhs, streams) if code >= 6 else
               CudaUnsignedRadixCiphertext(CudaVec(w, streams), 1, 1, lhs.lwe_dimension))
        _lib().scratch_cuda_integer_comparison_64_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), L, self.message_modulus, self.carry_modulus, code,
            False, True, self._noise_reduction())
        _lib().cuda_integer_comparison_64_async(s, C.byref(out._ffi()), C.byref(lhs._ffi()), C.byref(rhs._ffi()), mem, bsks,
                                                ksks)
        _lib().cleanup_cuda_integer_comparison_64(s, C.byref(mem))
        return out

    def scalar_compare(self, ct, scalar, op, streams):
        """Unsigned comparison of ONE integer with a clear scalar below 2^bits (radix/scalar_comparison.rs unchecked_scalar_*):
        the scalar's blocks stop at its last non-zero one, as BlockDecomposer::with_early_stop_at_zero yields them."""
        assert ct.num_integers == 1 and 0 <= int(scalar) < self.message_modulus ** ct.num_blocks
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        code = self.COMPARISONS[op]
        blocks, v = [], int(scalar)
        while v:
            blocks.append(v % self.message_modulus)
            v //= self.message_modulus
        h = np.ascontiguousarray(np.asarray(blocks, dtype=np.uint64))
        d = CudaVec(max(1, h.size), streams)
        if h.size:
            d.copy_from_cpu_async(h, streams)
        L, w = ct.num_blocks, ct.lwe_dimension + 1
        out = (CudaUnsignedRadixCiphertext.zeros_like(ct, streams) if code >= 6 else
               CudaUnsignedRadixCiphertext(CudaVec(w, streams), 1, 1, ct.lwe_dimension))
        _lib().scratch_cuda_integer_scalar_comparison_64_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), L, self.message_modulus, self.carry_modulus, code,
            False, True, self._noise_reduction())
        _lib().cuda_integer_scalar_comparison_64_async(s, C.byref(out._ffi()), C.byref(ct._ffi()), d.ptr,
                                                       h.ctypes.data_as(C.c_void_p), mem, bsks, ksks, h.size)
        _lib().cleanup_cuda_integer_scalar_comparison_64(s, C.byref(mem))
        streams.synchronize()
        return out

    def if_then_else(self, condition, ct_true, ct_false, streams):
        """condition ? ct_true : ct_false, ONE integer (radix/cmux.rs unchecked_if_then_else); condition: a boolean block."""
        assert ct_true.total_blocks == ct_false.total_blocks and condition.total_blocks >= 1
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        out = CudaUnsignedRadixCiphertext.zeros_like(ct_true, streams)
        _lib().scratch_cuda_cmux_64_async(s, C.byref(mem), self._bsk_params(), self._ksk_params(), ct_true.total_blocks,
                                          self.message_modulus, self.carry_modulus, True, self._noise_reduction())
        _lib().cuda_cmux_64_async(s, C.byref(out._ffi()), C.byref(condition._ffi()), C.byref(ct_true._ffi()),
                                  C.byref(ct_false._ffi()), mem, bsks, ksks)
        _lib().cleanup_cuda_cmux_64(s, C.byref(mem))
        return out

    def scalar_shift_assign(self, ct, shift, streams, left=True):
        """ct <<= shift / ct >>= shift (logical, clear amount; radix/scalar_shift.rs unchecked_scalar_{left,right}_shift_assign),
        ONE integer."""
        assert ct.num_integers == 1
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        _lib().scratch_cuda_logical_scalar_shift_64_inplace_async(
            s, C.byref(mem), self._bsk_params(), self._ksk_params(), ct.total_blocks, self.message_modulus,
            self.carry_modulus, 0 if left else 1, True, self._noise_reduction())
        _lib().cuda_logical_scalar_shift_64_inplace_async(s, C.byref(ct._ffi()), int(shift), mem, bsks, ksks)
        _lib().cleanup_cuda_logical_scalar_shift_64_inplace(s, C.byref(mem))

    def mul_assign(self, lhs, rhs, streams, return_pbs_count=False):
        """lhs *= rhs (mod 2^bits) on clean operands: schoolbook block products, column sums, propagation."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        _lib().hip_integer_scratch_batch(lhs.num_integers)
        _lib().scratch_cuda_integer_mult_inplace_64_async(
            s, C.byref(mem), False, False, self.message_modulus, self.carry_modulus, self._bsk_params(),
            self._ksk_params(), lhs.num_blocks, True, self._noise_reduction())
        pbs = int(_lib().hip_integer_mult_pbs_count(mem))
        _lib().cuda_integer_mult_inplace_64_async(s, C.byref(lhs._ffi()), False, C.byref(rhs._ffi()), False, bsks, ksks,
                                                  mem, self.bootstrapping_key.polynomial_size, lhs.num_blocks)
        _lib().cleanup_cuda_integer_mult_inplace_64(s, C.byref(mem))
        return pbs if return_pbs_count else None


    def mul_by_boolean_assign(self, ct, boolean, streams, boolean_is_left=False):
        """ct <- boolean ? ct : 0, block by block (integer_mult with is_boolean_right; with boolean_is_left the
        roles of the FFI operands are swapped: the boolean sits in the in/out operand, whose blocks receive the
        result — cuda/src/integer/multiplication.cuh:508-520).  `boolean`: one block per integer, value 0 or 1."""
        s, keep = self._streams(streams)
        ksks, bsks = self._key_ptrs(streams)
        mem = C.c_void_p()
        assert boolean.total_blocks == ct.num_integers
        _lib().hip_integer_scratch_batch(ct.num_integers)
        _lib().scratch_cuda_integer_mult_inplace_64_async(
            s, C.byref(mem), boolean_is_left, not boolean_is_left, self.message_modulus, self.carry_modulus,
            self._bsk_params(), self._ksk_params(), ct.num_blocks, True, self._noise_reduction())
        if boolean_is_left:
            # in/out operand: full-width buffer whose first `num_integers` blocks hold the booleans
            out = CudaUnsignedRadixCiphertext.zeros_like(ct, streams)
            w = ct.lwe_dimension + 1
            _lib().cuda_memcpy_async_gpu_to_gpu(out.d_blocks.ptr, boolean.d_blocks.ptr, boolean.total_blocks * w * 8,
                                                streams.ptr[0], streams.gpu_indexes[0])
            _lib().cuda_integer_mult_inplace_64_async(s, C.byref(out._ffi()), True, C.byref(ct._ffi()), False, bsks,
                                                      ksks, mem, self.bootstrapping_key.polynomial_size, ct.num_blocks)
            _lib().cleanup_cuda_integer_mult_inplace_64(s, C.byref(mem))
            return out
        _lib().cuda_integer_mult_inplace_64_async(s, C.byref(ct._ffi()), False, C.byref(boolean._ffi()), True, bsks,
                                                  ksks, mem, self.bootstrapping_key.polynomial_size, ct.num_blocks)
        _lib().cleanup_cuda_integer_mult_inplace_64(s, C.byref(mem))
        return ct


class CudaUnsignedRadixCiphertext:
    """A batch of unsigned radix integers on the device: [integer][block][lwe_size] u64, least
    significant block first (integer/gpu/ciphertext/mod.rs; one reference ciphertext = batch of 1)."""

    def __init__(self, d_blocks: CudaVec, num_integers, num_blocks, lwe_dimension):
        self.d_blocks = d_blocks
        self.num_integers, self.num_blocks, self.lwe_dimension = int(num_integers), int(num_blocks), int(lwe_dimension)
        # degrees / noise levels as the reference's structs carry them; the backend updates them and refuses operands
        # whose degrees exceed what an operation accepts.  Default 1: "not tracked" (callers that track set_degrees)
        self._info = np.ones(self.total_blocks, dtype=U64), np.ones(self.total_blocks, dtype=U64)

    @property
    def total_blocks(self):
        return self.num_integers * self.num_blocks

    def set_degrees(self, degree):
        self._info[0][:] = int(degree)

    @property
    def degrees(self):
        return self._info[0]

    @classmethod
    def from_blocks(cls, h_blocks, streams):
        """h_blocks: [integer][block][lwe_size] ciphertext words produced by the client key."""
        h = np.ascontiguousarray(h_blocks, dtype=U64)
        assert h.ndim == 3
        return cls(CudaVec.from_cpu_async(h.reshape(-1), streams), h.shape[0], h.shape[1], h.shape[2] - 1)

    @classmethod
    def zeros_like(cls, other, streams):
        return cls(CudaVec(other.total_blocks * (other.lwe_dimension + 1), streams), other.num_integers,
                   other.num_blocks, other.lwe_dimension)

    def duplicate(self, streams):
        h = self.to_blocks(streams)
        return CudaUnsignedRadixCiphertext.from_blocks(h, streams)

    def to_blocks(self, streams):
        return self.d_blocks.copy_to_cpu(streams).reshape(self.num_integers, self.num_blocks, self.lwe_dimension + 1)

    def _ffi(self):
        deg, noise = self._info
        return ffi.CudaRadixCiphertextFFI(self.d_blocks.ptr, deg.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          noise.ctypes.data_as(C.POINTER(C.c_uint64)), self.total_blocks,
                                          self.total_blocks, self.lwe_dimension)
