#!/usr/bin/env python3
"""The multi-GPU step of bench.py with the real RCCL backend on ONE GPU (world size 1): process group over "nccl", render
into a torch tensor on a side stream, all-reduce that tensor in place, barrier. Checks that the reduced image equals the
rendered one and that nothing hangs -- the part of the N-GPU path a one-GPU box can exercise with the real backend."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    scene = SceneData("cornell", 640, 360)
    t = Tracer(0)
    stream = torch.cuda.Stream(device=device)
    t.set_stream(stream.cuda_stream)
    scene.upload(t)
    image = torch.zeros((scene.height, scene.width, 4), dtype=torch.float32, device=device)
    t.target_bind_external(image.data_ptr(), image.data_ptr(), scene.width, scene.height)
    t.set_accum_mode(capi.ACCUM_SUM)
    t0 = time.perf_counter()
    for step in range(5):
        with torch.cuda.stream(stream):
            image.zero_()
            t.dispatch(1, 32, 1)
            before = image.clone()
            dist.all_reduce(image, op=dist.ReduceOp.SUM)   # what pathtracer_amd.dist does for world sizes above 1
            image.mul_(1.0 / 32.0)
        stream.synchronize()
        torch.cuda.synchronize(device)
        dist.barrier()
        assert torch.equal(image, before * (1.0 / 32.0)), "all-reduce over one rank changed the image"
    print(f"ok: 5 steps of render + RCCL all-reduce + barrier on one rank in {time.perf_counter() - t0:.3f} s")
    # the overlapped form bench.py uses: the exchange of frame k is in flight while frame k + 1 renders
    from pathtracer_amd import dist as ptdist
    merger = ptdist.OverlappedSampleMerge(image, 32, force_async=True)
    assert merger.asynchronous
    expected = []
    t1 = time.perf_counter()
    for step in range(6):
        with torch.cuda.stream(stream):
            image.zero_()
            t.dispatch(1 + step, 32, 1)          # another set of samples every frame, so frames differ
            expected.append(image.clone())
            if step > 0:
                got = merger.finish().clone()    # frame step - 1, merged
                want = expected[step - 1] * (1.0 / 32.0)
                want[..., 3] = 1.0
                stream.synchronize()
                assert torch.equal(got, want), f"overlapped merge of frame {step - 1} differs"
            merger.submit(image)
    with torch.cuda.stream(stream):
        last = merger.finish()
    stream.synchronize()
    torch.cuda.synchronize(device)
    dist.barrier()
    want = expected[-1] * (1.0 / 32.0)
    want[..., 3] = 1.0
    assert torch.equal(last, want), "overlapped merge of the last frame differs"
    print(f"ok: 6 frames with the exchange overlapped in {time.perf_counter() - t1:.3f} s")
    # the tile partition's merge (bench.py --partition tile): all-gather of the owned rows, also beside the next frame
    t.set_accum_mode(capi.ACCUM_MEAN)
    gather = ptdist.TileGather(image, 0, 1, 16, overlapped=True, force_exchange=True)
    assert gather.asynchronous
    frames = []
    t2 = time.perf_counter()
    for step in range(5):
        with torch.cuda.stream(stream):
            image.zero_()
            image[..., 3] = 1.0
            t.dispatch(1, 8 + step, 1)           # another sample count every frame, so frames differ
            frames.append(image.clone())
            if step > 0:
                got = gather.finish().clone()
                stream.synchronize()
                assert torch.equal(got, frames[step - 1]), f"gathered frame {step - 1} differs"
            gather.submit(image)
    with torch.cuda.stream(stream):
        last = gather.finish()
    stream.synchronize()
    torch.cuda.synchronize(device)
    dist.barrier()
    assert torch.equal(last, frames[-1]), "gathered last frame differs"
    print(f"ok: 5 frames through the RCCL all-gather of the owned rows in {time.perf_counter() - t2:.3f} s")
    t.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
