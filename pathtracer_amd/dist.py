"""Multi-GPU partitions of the accumulation (one process per GPU, torch.distributed; backend "nccl" is RCCL over
xGMI on the GPU box, "gloo" in CPU tests).

Every (pixel, sample) is independent and the shader's random numbers are a pure function of pixel centre and sample
index, so the work shards two ways with no exchange until the end:

  sample split   rank r renders sample indices r+1, r+1+R, r+1+2R, ... with the TRUE index in the RNG and keeps a
                 running fp32 SUM; one all-reduce(sum) of the image merges them and a scale by 1/N gives the mean.
                 Differs from the sequential running mean only by fp32 rounding.
  tile split     rank r owns a set of rows (interleaved blocks, for balance) and runs ALL samples for them with the
                 reference recurrence, so its pixels are bit-identical to a 1-GPU render; the merge is an all-gather of
                 the owned rows (bytes only, no arithmetic), which can run beside the rendering of the next frame.

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


class TileGather:
    """Tile-split merge: every rank sends the rows it owns and receives everyone else's -- an all-gather, half the bytes
    of a sum all-reduce over the zero-padded image and no arithmetic, so owned pixels arrive bit for bit.

    Rank r owns the row blocks r, r + R, ... (`row_blocks`). Its rows are packed into one contiguous send buffer, padded
    to the largest per-rank row count (heights that are not a multiple of block x world give ranks different counts), all
    buffers are gathered with one `all_gather_into_tensor`, and the valid rows are scattered to their places in the merged
    image. `submit(image)` packs and starts the exchange without waiting (RCCL runs it on its own stream, ordered after
    the pack), so the caller may clear `image` and render the next frame at once; `finish()` waits for the exchange in
    flight and returns the merged image; `submit` calls it first, so one set of buffers is enough. With
    `overlapped=False`, or over a backend that cannot gather device memory (gloo rehearsals on a GPU stage through the
    host), `submit` completes the exchange itself. Call both under the stream the renderer uses."""

    def __init__(self, like, rank, world, block=16, group=None, overlapped=True, force_exchange=False):
        H = like.shape[0]
        rows = [torch.tensor([y for y0, h in row_blocks(r, world, H, block) for y in range(y0, y0 + h)], dtype=torch.long) for r in range(world)]
        self.max_rows = max(1, max(len(r) for r in rows))
        self.my_rows = rows[rank].to(like.device)
        self.n_mine = len(rows[rank])
        # position in the gathered buffer -> image row, for every valid (non-padding) row
        self.src = torch.cat([torch.arange(len(r), dtype=torch.long) + k * self.max_rows for k, r in enumerate(rows)]).to(like.device)
        self.dst = torch.cat(rows).to(like.device)
        self.send = torch.zeros((self.max_rows,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        self.recv = torch.zeros((world * self.max_rows,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        self.merged = torch.empty_like(like)
        self.group = group
        self.world = world
        self.work = None
        self.pending = False
        # (force_exchange: a one-rank group still goes through the backend's all-gather, for tests of the call sequence)
        on = dist.is_available() and dist.is_initialized() and (world > 1 or force_exchange)
        self.force_exchange = bool(force_exchange)
        self.direct = on and (not like.is_cuda or dist.get_backend(group) == "nccl")
        self.staged = on and not self.direct
        self.asynchronous = bool(overlapped and self.direct)
        self.bytes_sent_per_rank = self.send.numel() * self.send.element_size() * max(0, world - 1)

    def submit(self, image):
        self.finish()
        if self.n_mine:
            torch.index_select(image, 0, self.my_rows, out=self.send[:self.n_mine])
        self.pending = True
        if (self.world <= 1 and not self.force_exchange) or not (self.direct or self.staged):
            self.recv[:self.max_rows].copy_(self.send)
        elif self.staged:
            host_send, host_recv = self.send.cpu(), torch.empty(self.recv.shape, dtype=self.recv.dtype)
            dist.all_gather_into_tensor(host_recv, host_send, group=self.group)
            self.recv.copy_(host_recv)
        else:
            self.work = dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=self.asynchronous)
        if not self.asynchronous:
            self.finish()

    def finish(self):
        if self.work is not None:
            self.work.wait()  # the current stream waits for the exchange; the host does not
            self.work = None
        if self.pending:
            self.merged.index_copy_(0, self.dst, self.recv.index_select(0, self.src))
            self.pending = False
        return self.merged


def merge_tile_split(image, height=None, block=16, group=None, rank=None, world=None):
    """One-shot form of TileGather: every rank ends up with the full image (in place). Rows a rank does not own are
    overwritten with their owner's; what they held before does not matter."""
    on = dist.is_available() and dist.is_initialized()
    rank = (dist.get_rank(group) if on else 0) if rank is None else rank
    world = (dist.get_world_size(group) if on else 1) if world is None else world
    g = TileGather(image, rank, world, block, group, overlapped=False)
    g.submit(image)
    image.copy_(g.finish())
    return image


def zero_unowned_rows(image, rank, world, block=16):
    own = torch.zeros(image.shape[0], dtype=torch.bool, device=image.device)
    for y0, h in row_blocks(rank, world, image.shape[0], block):
        own[y0:y0 + h] = True
    image[~own] = 0
    return image
