"""Samples-per-pixel sharding across the GPUs of a node: one process per GPU, `torch.distributed` (backend "nccl",
which is RCCL on ROCm) for the single collective of the path -- a sum all-reduce of the f64 film over xGMI.

The reference's own precedent is thrender_no_rayon (camera.rs:319-328), which gives each of 16 threads spp/16 samples
of the whole frame (and never merges them).  Here rank r of R renders global sample indices
[begin_r, begin_r + count_r); the random stream of a sample is keyed by (pixel, global sample index), so the union
of the shards is the same set of paths whatever R is, and the summed film differs from a single-GPU film only by
the association of the per-shard partial sums."""
import torch
import torch.distributed as dist


def shard_range(spp_total, rank, world):
    """Contiguous sample range of `rank`; the remainder goes to the low ranks (nothing is dropped, unlike
    camera.rs:324 which truncates spp/16)."""
    base, rem = divmod(spp_total, world)
    count = base + (1 if rank < rem else 0)
    begin = rank * base + min(rank, rem)
    return begin, count


def render_sharded(render_range, film, spp_total, group=None):
    """render_range(begin, count) must fill `film` (a torch f64 tensor of shape (H, W, 3), on the GPU for nccl) with
    the SUM of the radiances of samples [begin, begin + count).  After the call every rank holds the whole-job film."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    begin, count = shard_range(spp_total, rank, world)
    if count > 0:
        render_range(begin, count)
    else:
        film.zero_()
    if world > 1:
        dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
        if film.is_cuda:
            # the renderer works on a stream of its own, ordered against this one only through the host: the sum
            # must have landed before the caller renders into the same film again (or reads it through the C ABI)
            torch.cuda.current_stream(film.device).synchronize()
    return begin, count


def render_scene_sharded(scene, cam, spp_total, max_depth, background, seed=42, batch_paths=0, group=None):
    """Renders `spp_total` samples per pixel of a committed Scene across all ranks; returns (film tensor on this
    rank's GPU, stats of this rank)."""
    film = torch.empty((cam.height, cam.width, 3), dtype=torch.float64, device=f"cuda:{scene.device}")
    stats = {}

    def render_range(begin, count):
        stats.update(scene.render_into(cam, count, max_depth, background, film.data_ptr(), seed=seed, spp_begin=begin,
                                       batch_paths=batch_paths))

    render_sharded(render_range, film, spp_total, group)
    return film, stats
