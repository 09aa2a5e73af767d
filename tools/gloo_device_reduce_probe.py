#!/usr/bin/env python3
"""Why did a gloo all_reduce of a DEVICE image not return in the two-ranks-on-one-GPU rehearsal? One bounded experiment.

Two ranks share GPU 0 (gloo process group, like the rehearsal of tests/test_bench_launch.py). Each rank runs, one after the other,
every step under a watchdog that dumps the Python stacks of all threads and ends the process when the step takes longer than
PT_PROBE_STEP_S seconds (default 25), so a hang names its step and cannot outlive the run:

  a  all_reduce of a device tensor made by torch alone, on the default stream
  b  the same inside `with torch.cuda.stream(side_stream)`
  c  the image pt_dispatch rendered into on that side stream (the stream handed to pt_set_stream), reduced inside the same
     `with` block right after the dispatch -- the call sequence of the rehearsal that hung
  d  as c, but after pt_finish (nothing of the renderer in flight)
  e  as c, through a host copy (what bench.py does for non-RCCL backends)
  f  the whole step loop of that rehearsal: three warm-up steps and eight timed ones issued back to back (256 samples each, the
     reduce of one step queued behind its dispatch), fences with a barrier, then the float64 timing tensors reduced on the device

Prints one JSON line per rank and step: seconds, and whether the result is the expected sum. The parent starts the ranks, relays
their output, and ends whatever is still alive after the overall limit."""
import faulthandler
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEP_S = float(os.environ.get("PT_PROBE_STEP_S", "25"))


def rank_main():
    import torch
    import torch.distributed as dist
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = torch.cuda.Stream(device=dev)
    scene = SceneData("cornell", 1920, 1080)
    tracer = Tracer(0)
    tracer.set_stream(side.cuda_stream)
    scene.upload(tracer)
    image = torch.zeros((scene.height, scene.width, 4), dtype=torch.float32, device=dev)
    tracer.target_bind_external(image.data_ptr(), image.data_ptr(), scene.width, scene.height)
    tracer.set_accum_mode(capi.ACCUM_SUM)

    def step(name, fn):
        faulthandler.dump_traceback_later(STEP_S, exit=True)
        t0 = time.perf_counter()
        ok = fn()
        torch.cuda.synchronize(dev)
        faulthandler.cancel_dump_traceback_later()
        print(json.dumps({"rank": rank, "step": name, "seconds": round(time.perf_counter() - t0, 3), "ok": bool(ok)}), flush=True)
        dist.barrier()

    def plain(stream=None):
        def run():
            t = torch.full((1080, 1920, 4), float(rank + 1), device=dev)
            dist.all_reduce(t)
            return bool((t == float(sum(range(1, world + 1)))).all().item())
        if stream is None:
            return run()
        with torch.cuda.stream(stream):
            return run()

    def rendered(mode):
        with torch.cuda.stream(side):
            image.zero_()
            tracer.dispatch(1 + rank, 64, world)
            if mode == "finish":
                tracer.finish()
            mine = image.clone()
            if mode == "host":
                host = image.cpu()
                dist.all_reduce(host)
                image.copy_(host)
            else:
                dist.all_reduce(image)
            # every rank's sums are positive where the box is seen: the reduced image is at least this rank's own
            return bool((image[..., :3] >= mine[..., :3]).all().item() and (image[..., :3] > mine[..., :3]).any().item())

    def old_loop():
        """The step loop of the rehearsal as it stood when the hang was seen: dispatch and direct reduce back to back without
        waiting, a fence with a barrier, then the timing tensors (float64, on the device) reduced with MAX and SUM."""
        def one():
            with torch.cuda.stream(side):
                image.zero_()
                tracer.dispatch(1 + rank, 256, world)
                dist.all_reduce(image)
                image.mul_(1.0 / 512.0)
                image[..., 3] = 1.0

        def fence():
            side.synchronize()
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
        tracer.enable_timing(True)
        for _ in range(3):
            one()
        fence()
        tracer.reset_stats()
        for _ in range(8):
            one()
        fence()
        st = tracer.stats()
        t = torch.tensor([1.0 + rank], dtype=torch.float64, device=dev)
        counts = torch.tensor([float(st["rays"]), float(st["samples"])], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        return float(t.item()) == float(world) and float(counts[1].item()) == 1920 * 1080 * 256 * 8 * world

    step("a_plain_default_stream", lambda: plain())
    step("b_plain_side_stream", lambda: plain(side))
    step("c_rendered_same_stream", lambda: rendered("direct"))
    step("d_rendered_after_finish", lambda: rendered("finish"))
    step("e_rendered_host_copy", lambda: rendered("host"))
    step("f_step_loop_as_it_hung", old_loop)
    tracer.close()
    dist.destroy_process_group()


def main():
    if "RANK" in os.environ:
        return rank_main()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", TORCH_DISTRIBUTED_DEBUG="DETAIL")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(env, RANK=str(r))) for r in range(2)]
    deadline = time.time() + 7 * STEP_S + 120
    while time.time() < deadline and any(p.poll() is None for p in procs):
        time.sleep(0.5)
        if any(p.poll() not in (None, 0) for p in procs):      # a rank's watchdog fired: the other one waits in vain
            time.sleep(STEP_S + 5)
            break
    for p in procs:
        if p.poll() is None:
            p.kill()
    codes = [p.wait() for p in procs]
    print(json.dumps({"exit_codes": codes}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
