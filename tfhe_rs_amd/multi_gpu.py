"""Batch sharding across the GPUs of one node — no collectives.

The reference distributes a batch of independent LWEs by contiguous chunks and replicates
BSK / KSK / LUTs per GPU (backends/tfhe-cuda-backend/cuda/src/utils/helper_multi_gpu.cu:71-101
`get_num_inputs_on_gpu`; tfhe/src/core_crypto/gpu/ffi.rs:744-787).  Here each rank (one
process per GPU) owns one contiguous shard; nothing is exchanged on the data path.
"""


def get_num_inputs_on_gpu(total_num_inputs: int, gpu_index: int, gpu_count: int) -> int:
    """ceil(B/G) on the first B mod G GPUs, floor(B/G) on the rest (helper_multi_gpu.cu:71-101)."""
    assert 0 <= gpu_index < gpu_count
    small = total_num_inputs // gpu_count
    remainder = total_num_inputs % gpu_count
    return small + 1 if gpu_index < remainder else small


def get_gpu_offset(total_num_inputs: int, gpu_index: int, gpu_count: int) -> int:
    """first LWE index owned by gpu_index (helper_multi_gpu.cu: get_gpu_offset)."""
    return sum(get_num_inputs_on_gpu(total_num_inputs, g, gpu_count) for g in range(gpu_index))


def shard_range(total_num_inputs: int, rank: int, world_size: int):
    start = get_gpu_offset(total_num_inputs, rank, world_size)
    return start, start + get_num_inputs_on_gpu(total_num_inputs, rank, world_size)
