"""Output-timeline sharding across ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards trivially: output frame k depends only on k, the configuration, the table and the input
frames within `integer_stretched_kernel_radius` of its position.  Rank r therefore runs an ordinary low-level call on its
slice of the input whose "padding" is the real neighbouring frames (which the reference allows, clownresampler.h:725-733)
from the closed-form state at its first output frame - ClownResamplerAMD_PlanShard gives all of that.  There is NO
collective in the data path; the only communication is the optional final concatenate of the int32 outputs
(`gather_output`: one all_gather of equal-sized, padded shards; a one-shot exchange over xGMI, not a ring reduction).
"""
import clownresampler_amd as cr


def shard_of(api, state, total_input_frames, rank, world_size):
    """ClownResamplerAMD_Shard for `rank`: output range, input pointer offset / frame count, halo, start state."""
    return api.PlanShard(state, total_input_frames, rank, world_size)


def resample_shard_device(api, plan, shard, device_input_ptr, device_output_ptr, hip_stream=None):
    """Enqueues this rank's shard.  device_input_ptr points at the shard's own padded slice: padded-buffer frame
    `shard.first_input_frame` of the whole stream (i.e. its halo starts `halo_frames` before its first input frame)."""
    st = cr.LowLevel_State.from_buffer_copy(shard.state)
    n, left, ran_out = api.ResampleDevice(plan, st, device_input_ptr, shard.input_frames, device_output_ptr, shard.output_frames, hip_stream)
    assert n == shard.output_frames
    return n


def gather_output(local_out, shard, total_output_frames, channels, world_size, group=None):
    """Concatenates the ranks' int32 outputs; every rank gets the whole stream (torch tensor on local_out's device).
    Shards are padded to the common per-rank frame count ceil(total / world) and the padding trimmed after the exchange."""
    import torch
    import torch.distributed as dist

    per = (total_output_frames + world_size - 1) // world_size
    send = torch.zeros(per * channels, dtype=torch.int32, device=local_out.device)
    send[: shard.output_frames * channels] = local_out[: shard.output_frames * channels]
    recv = torch.empty(per * channels * world_size, dtype=torch.int32, device=local_out.device)
    if world_size == 1:
        recv.copy_(send)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    return recv[: total_output_frames * channels]


def gather_output_to_root(local_out, shard, total_output_frames, channels, world_size, root=0, group=None):
    """Concatenates the ranks' int32 outputs on `root` only (ncclGather semantics: with RCCL every peer -> root transfer is a
    point-to-point send over its own xGMI link).  Returns the whole stream on root, None elsewhere."""
    import torch
    import torch.distributed as dist

    per = (total_output_frames + world_size - 1) // world_size
    send = torch.zeros(per * channels, dtype=torch.int32, device=local_out.device)
    send[: shard.output_frames * channels] = local_out[: shard.output_frames * channels]
    if world_size == 1:
        return send[: total_output_frames * channels]
    rank = dist.get_rank(group)
    recv = torch.empty(per * channels * world_size, dtype=torch.int32, device=local_out.device) if rank == root else None
    dist.gather(send, list(recv.split(per * channels)) if rank == root else None, dst=root, group=group)
    return recv[: total_output_frames * channels] if rank == root else None
