"""Multi-GPU partitions of the accumulation (one process per GPU, torch.distributed; backend "nccl" is RCCL over
xGMI on the GPU box, "gloo" in CPU tests).

Every (pixel, sample) is independent and the shader's random numbers are a pure function of pixel centre and sample
index, so the work shards two ways with no exchange until the end:

  sample split   rank r renders sample indices r+1, r+1+R, r+1+2R, ... with the TRUE index in the RNG and keeps a
                 running fp32 SUM; one all-reduce(sum) of the image merges them and a scale by 1/N gives the mean.
                 Differs from the sequential running mean only by fp32 rounding.
  tile split     rank r owns a set of rows (interleaved blocks, for balance) and runs ALL samples for them with the
                 reference recurrence, so its pixels are bit-identical to a 1-GPU render; the merge moves bytes only.

The renderer is injected (`render_sum(first, stride, count)` / `render_rows(y0, h)`), so this module holds no
tracing code: on the GPU box the callables wrap pathtracer_amd.Tracer, in CPU tests they wrap a checker.
"""
import torch
import torch.distributed as dist


def sample_indices(rank, world, total_samples):
    """(first_sample, stride, count) of the 1-based sample indices rank `rank` renders out of 1..total_samples."""
    if total_samples < 0 or not (0 <= rank < world):
        raise ValueError("bad partition arguments")
    count = (total_samples - rank + world - 1) // world if total_samples > rank else 0
    return rank + 1, world, count


def row_blocks(rank, world, height, block=16):
    """Row ranges [(y0, h), ...] of rank `rank`: blocks of `block` rows dealt round-robin, so expensive image regions
    are shared out. The union over ranks covers 0..height exactly once."""
    if height < 0 or block <= 0 or not (0 <= rank < world):
        raise ValueError("bad partition arguments")
    out = []
    for i, y0 in enumerate(range(0, height, block)):
        if i % world == rank:
            out.append((y0, min(block, height - y0)))
    return out


def _all_reduce_sum(t, group=None):
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    if t.is_cuda and dist.get_backend(group) != "nccl":
        # CPU-side backends (gloo rehearsals on a one-GPU box): stage through host memory
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)  # RCCL over xGMI on the GPU box


def merge_sample_split(sum_image, total_samples, group=None):
    """In-place all-reduce of the per-rank running sums, then scale to the mean. `sum_image`: [H, W, 4] fp32 tensor
    whose rgb holds this rank's sum; alpha is overwritten with 1."""
    _all_reduce_sum(sum_image, group)
    sum_image.mul_(1.0 / float(total_samples))
    sum_image[..., 3] = 1.0
    return sum_image


class OverlappedSampleMerge:
    """Sample-split merge that runs beside the rendering of the NEXT frame.

    `submit(sum_image)` copies this rank's running sums into a staging image and starts the all-reduce of that copy
    without waiting for it (RCCL runs it on its own stream, ordered after the copy), so the caller can clear
    `sum_image` and render the next frame at once; the exchange of one frame (a few hundred microseconds for a 1080p
    RGBA32F image over xGMI) hides behind the milliseconds of the next. `finish()` waits for the exchange in flight,
    scales the sums to the mean and returns the merged image; `submit` calls it first, so one staging image is enough.
    Device images over a backend that reduces on the host (gloo rehearsals on a GPU) are served synchronously. Call both under the stream the
    renderer uses."""

    def __init__(self, like, total_samples, group=None, force_async=False):
        self.merged = torch.empty_like(like)
        self.total_samples = float(total_samples)
        self.group = group
        self.work = None
        self.scaled = True
        on = dist.is_available() and dist.is_initialized()
        # device images need a backend that reduces device memory (RCCL); host images go through any backend
        direct = on and (not like.is_cuda or dist.get_backend(group) == "nccl")
        self.asynchronous = bool(direct and (dist.get_world_size(group) > 1 or force_async))

    def _scale(self):
        if not self.scaled:
            self.merged.mul_(1.0 / self.total_samples)
            self.merged[..., 3] = 1.0
            self.scaled = True

    def submit(self, sum_image):
        self.finish()
        self.merged.copy_(sum_image)
        self.scaled = False
        if self.asynchronous:
            self.work = dist.all_reduce(self.merged, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            _all_reduce_sum(self.merged, self.group)
            self._scale()

    def finish(self):
        if self.work is not None:
            self.work.wait()  # the current stream waits for the exchange; the host does not
            self.work = None
        self._scale()
        return self.merged


def merge_tile_split(image, height, block=16, group=None):
    """Every rank ends up with the full image: rows it does not own are zero before the call, so a sum all-reduce
    moves each row from its owner to everyone without changing a bit (x + 0 + ... + 0 is exact)."""
    _all_reduce_sum(image, group)
    return image


def zero_unowned_rows(image, rank, world, block=16):
    own = torch.zeros(image.shape[0], dtype=torch.bool, device=image.device)
    for y0, h in row_blocks(rank, world, image.shape[0], block):
        own[y0:y0 + h] = True
    image[~own] = 0
    return image
