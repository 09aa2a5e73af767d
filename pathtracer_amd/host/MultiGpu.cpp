// The render loop on several GPUs of one node: one process, one context + one host thread + a render and an exchange stream
// per rank, images merged over RCCL (librccl is opened at run time, so a host without it still renders on one GPU).
// The reference has one GL context and one frame in flight (PathTracer.cpp:1301-1500); every (pixel, sample) of its shader is
// independent and its random numbers are a pure function of (pixel centre, sample index) (pathtracing.fs:27-36), so the frame
// loop shards with no exchange until an image is wanted:
//   sample partition  rank r renders samples r + 1, r + 1 + N, ... of the whole frame into a running SUM (PT_ACCUM_SUM);
//                     ncclAllReduce(ncclFloat, ncclSum) of a copy of the sums, scaled by 1 / samples at the end;
//   tile partition    rank r renders every sample of the rows in its interleaved blocks (pt_set_row_interleave) with the
//                     reference recurrence; ncclAllGather of the packed rows, unpacked into the merged image -- no arithmetic,
//                     the merged image equals a one-GPU render bit for bit.
// The exchange of frame k runs on its own stream beside the rendering of frame k + 1.
#include <dlfcn.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "RenderLoop.h"
#include "Scene.h"
#include "Shader.h"
#include "Workloads.h"
#include "pt_hip.h"

namespace pthost {

namespace {

// ---- RCCL, opened at run time ---------------------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    bool open(std::string* err) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            if (err) *err = std::string("cannot open librccl.so: ") + dlerror();
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(sym("ncclCommAbort"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !CommAbort || !AllReduce || !AllGather || !GetErrorString) {
            if (err) *err = "librccl.so lacks an entry point this host needs";
            return false;
        }
        return true;
    }
    // (the library stays open for the life of the process: closing it while its proxy threads wind down is not worth the risk)
};

// ---- a barrier for the rank threads that can be broken when a rank fails -------------------------------------------------
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    // false when the barrier was broken (some rank failed): the caller stops
    bool wait() {
        std::unique_lock<std::mutex> lock(m_);
        if (broken_) return false;
        const unsigned gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            ++gen_;
            cv_.notify_all();
            return true;
        }
        cv_.wait(lock, [&] { return gen_ != gen || broken_; });
        return !broken_;
    }
    void break_all() {
        std::lock_guard<std::mutex> lock(m_);
        broken_ = true;
        cv_.notify_all();
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    unsigned gen_ = 0;
    bool broken_ = false;
};

// Interleaved row blocks of the tile partition: block b = rows [b * B, min(H, (b + 1) * B)), owned by rank b % N. In a rank's
// packed buffer its blocks follow each other, `block_floats` apart (the ragged last block of the image is shorter).
struct RowLayout {
    int width = 0, height = 0, block_rows = 16, ranks = 1;
    int blocks() const { return (height + block_rows - 1) / block_rows; }
    int blocks_of(int rank) const { return blocks() > rank ? (blocks() - rank + ranks - 1) / ranks : 0; }
    size_t block_floats() const { return (size_t)block_rows * width * 4; }
    size_t packed_floats() const { return (size_t)blocks_of(0) * block_floats(); }   // rank 0 owns the most blocks
    int rows_of_block(int b) const { return std::min(block_rows, height - b * block_rows); }
};

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        const hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return std::string(#expr) + " failed: " + hipGetErrorString(e_);          \
    } while (0)

// image (full frame on the device) -> packed (this rank's row blocks), or the other way round, on `stream`. Full blocks of a
// rank lie `ranks x block` apart in the image and one block apart in the packed buffer: one strided copy; the ragged last
// block of the image, if this rank owns it, is one more.
std::string copy_blocks(const RowLayout& lay, int rank, float* image, float* packed, bool pack, hipStream_t stream) {
    const int owned = lay.blocks_of(rank);
    if (owned == 0) return "";
    const int last = rank + (owned - 1) * lay.ranks;   // this rank's last block
    const bool ragged = lay.rows_of_block(last) != lay.block_rows;
    const int full = ragged ? owned - 1 : owned;
    const size_t block_bytes = lay.block_floats() * sizeof(float);
    float* first = image + (size_t)rank * lay.block_floats();
    if (full > 0) {
        if (pack) HIP_TRY(hipMemcpy2DAsync(packed, block_bytes, first, block_bytes * lay.ranks, block_bytes, (size_t)full, hipMemcpyDeviceToDevice, stream));
        else HIP_TRY(hipMemcpy2DAsync(first, block_bytes * lay.ranks, packed, block_bytes, block_bytes, (size_t)full, hipMemcpyDeviceToDevice, stream));
    }
    if (ragged) {
        const size_t bytes = (size_t)lay.rows_of_block(last) * lay.width * 16;
        float* in_image = image + (size_t)last * lay.block_floats();
        float* in_packed = packed + (size_t)full * lay.block_floats();
        HIP_TRY(hipMemcpyAsync(pack ? in_packed : in_image, pack ? in_image : in_packed, bytes, hipMemcpyDeviceToDevice, stream));
    }
    return "";
}

struct Shared {
    RenderOptions opt;
    MultiGpuOptions multi;
    const Workload* workload = nullptr;
    int ranks = 0;
    bool use_rccl = false;
    Rccl* rccl = nullptr;
    std::vector<ncclComm_t> comms;
    Barrier* barrier = nullptr;
    RowLayout layout;
    size_t image_floats = 0;
    // host exchange: where the ranks' buffers meet
    std::vector<std::vector<float>> host_parts;
    std::vector<float> host_merged;
    // results
    std::mutex m;
    std::string error;
    std::atomic<bool> failed{false};
    bool comms_aborted = false;   // (under m)
    std::vector<PtStats> stats;
    std::vector<float> final_image;   // rank 0's merged image (sample partition: sums)
};

struct RankState {
    int rank = 0, device = 0;
    PtContext* ctx = nullptr;
    hipStream_t render = nullptr, comm = nullptr;
    hipEvent_t copied = nullptr, merged_done = nullptr;
    float* staging = nullptr;   // sample partition: copy of the sums, reduced in place. tile partition: packed rows of this rank
    float* gathered = nullptr;  // tile partition: every rank's packed rows
    float* merged = nullptr;    // tile partition: the whole frame
    bool exchange_pending = false;
};

void fail(Shared& sh, int rank, const std::string& what) {
    {
        std::lock_guard<std::mutex> lock(sh.m);
        if (sh.error.empty()) sh.error = "rank " + std::to_string(rank) + ": " + what;
    }
    if (!sh.failed.exchange(true)) {
        sh.barrier->break_all();
        // ranks already inside a collective would wait for this one for ever
        if (sh.use_rccl) {
            std::lock_guard<std::mutex> lock(sh.m);
            for (ncclComm_t c : sh.comms)
                if (c) (void)sh.rccl->CommAbort(c);
            sh.comms_aborted = true;
        }
    }
}

// Merges what the ranks hold after a frame. Sample partition: staging <- sum over ranks of staging. Tile partition: merged <-
// all rows. With RCCL everything is queued on the exchange stream behind `copied`; the host exchange is synchronous.
std::string exchange(Shared& sh, RankState& rs) {
    const bool tile = sh.multi.partition == MultiGpuOptions::TILE;
    const size_t count = tile ? sh.layout.packed_floats() : sh.image_floats;
    if (sh.use_rccl) {
        HIP_TRY(hipStreamWaitEvent(rs.comm, rs.copied, 0));
        ncclResult_t r;
        if (tile) r = sh.rccl->AllGather(rs.staging, rs.gathered, count, ncclFloat, sh.comms[rs.rank], rs.comm);
        else r = sh.rccl->AllReduce(rs.staging, rs.staging, count, ncclFloat, ncclSum, sh.comms[rs.rank], rs.comm);
        if (r != ncclSuccess) return std::string("RCCL: ") + sh.rccl->GetErrorString(r);
    } else {
        // the buffers meet in host memory: every rank brings its own, rank 0 combines, every rank takes the result
        HIP_TRY(hipStreamWaitEvent(rs.comm, rs.copied, 0));
        HIP_TRY(hipMemcpyAsync(sh.host_parts[rs.rank].data(), rs.staging, count * sizeof(float), hipMemcpyDeviceToHost, rs.comm));
        HIP_TRY(hipStreamSynchronize(rs.comm));
        if (!sh.barrier->wait()) return "another rank failed";
        if (rs.rank == 0) {
            if (tile) {
                for (int q = 0; q < sh.ranks; ++q) memcpy(sh.host_merged.data() + (size_t)q * count, sh.host_parts[q].data(), count * sizeof(float));
            } else {
                // fp32 sums in rank order (RCCL's ring adds in another order: both differ from a sequential mean by rounding only)
                std::copy(sh.host_parts[0].begin(), sh.host_parts[0].end(), sh.host_merged.begin());
                for (int q = 1; q < sh.ranks; ++q)
                    for (size_t i = 0; i < count; ++i) sh.host_merged[i] += sh.host_parts[q][i];
            }
        }
        if (!sh.barrier->wait()) return "another rank failed";
        HIP_TRY(hipMemcpyAsync(tile ? rs.gathered : rs.staging, sh.host_merged.data(), (tile ? count * sh.ranks : count) * sizeof(float),
                               hipMemcpyHostToDevice, rs.comm));
    }
    if (tile) {
        for (int q = 0; q < sh.ranks; ++q) {
            const std::string e = copy_blocks(sh.layout, q, rs.merged, rs.gathered + (size_t)q * count, false, rs.comm);
            if (!e.empty()) return e;
        }
    }
    HIP_TRY(hipEventRecord(rs.merged_done, rs.comm));
    if (!sh.use_rccl) {
        HIP_TRY(hipStreamSynchronize(rs.comm));            // host_merged is rewritten by the next frame
        if (!sh.barrier->wait()) return "another rank failed";
    }
    rs.exchange_pending = true;
    return "";
}

std::string rank_body(Shared& sh, RankState& rs) {
#define PT_TRY(call)                                                       \
    do {                                                                   \
        if ((call) != PT_OK) return std::string(pt_last_error());          \
    } while (0)
    const RenderOptions& opt = sh.opt;
    const bool tile = sh.multi.partition == MultiGpuOptions::TILE;
    const int N = sh.ranks;
    HIP_TRY(hipSetDevice(rs.device));
    HIP_TRY(hipStreamCreateWithFlags(&rs.render, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&rs.comm, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&rs.copied, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&rs.merged_done, hipEventDisableTiming));
    PT_TRY(pt_create(rs.device, &rs.ctx));
    PT_TRY(pt_set_stream(rs.ctx, rs.render));
    uint32_t program = 0;
    std::string err;
    if (!SetupContext(rs.ctx, opt, *sh.workload, &program, &err)) return err;
    if (tile) {
        PT_TRY(pt_set_row_interleave(rs.ctx, sh.multi.block_rows, rs.rank, N));
        HIP_TRY(hipMalloc(&rs.staging, sh.layout.packed_floats() * sizeof(float)));
        HIP_TRY(hipMalloc(&rs.gathered, sh.layout.packed_floats() * N * sizeof(float)));
        HIP_TRY(hipMalloc(&rs.merged, sh.image_floats * sizeof(float)));
        // (blocks a rank lacks at the end of its packed buffer travel as they are: nothing reads them)
        HIP_TRY(hipMemsetAsync(rs.staging, 0, sh.layout.packed_floats() * sizeof(float), rs.render));
    } else {
        PT_TRY(pt_set_accum_mode(rs.ctx, PT_ACCUM_SUM));
        PT_TRY(pt_target_clear(rs.ctx));   // sums start at (0, 0, 0, 0)
        HIP_TRY(hipMalloc(&rs.staging, sh.image_floats * sizeof(float)));
    }
    if (!sh.barrier->wait()) return "another rank failed";   // every context stands before any collective is queued

    // sample indices of this rank: tile partition all of them, sample partition its residue class
    const int total = opt.passes;
    const int mine = tile ? total : (total > rs.rank ? (total - rs.rank + N - 1) / N : 0);
    const int longest = tile ? total : (total + N - 1) / N;   // rank 0's count: every rank runs that many frames' worth of exchanges
    const int batch = opt.samples_per_dispatch > 0 ? opt.samples_per_dispatch : 1;
    const int frames = (longest + batch - 1) / batch;
    int done = 0;
    for (int f = 0; f < frames; ++f) {
        const int count = std::min(batch, mine - done);
        // PathTracer.cpp:1370-1403
        PT_TRY(pt_use_program(rs.ctx, program));
        if (!SetFrameUniforms(rs.ctx, opt, *sh.workload)) return std::string(pt_last_error());
        if (count > 0) {
            if (tile) PT_TRY(pt_dispatch(rs.ctx, 1 + done, count, 1));
            else PT_TRY(pt_dispatch(rs.ctx, 1 + rs.rank + done * N, count, N));
            done += count;
        }
        const bool last = f + 1 == frames;
        if (!sh.multi.exchange_every_frame && !last) continue;
        // a copy of what this rank holds now goes to the exchange; the render stream goes on with the next frame
        void* image = nullptr;
        PT_TRY(pt_current_image(rs.ctx, &image));
        if (rs.exchange_pending) HIP_TRY(hipStreamWaitEvent(rs.render, rs.merged_done, 0));   // the buffers of the previous exchange are free
        if (tile) {
            const std::string e = copy_blocks(sh.layout, rs.rank, static_cast<float*>(image), rs.staging, true, rs.render);
            if (!e.empty()) return e;
        } else {
            HIP_TRY(hipMemcpyAsync(rs.staging, image, sh.image_floats * sizeof(float), hipMemcpyDeviceToDevice, rs.render));
        }
        HIP_TRY(hipEventRecord(rs.copied, rs.render));
        const std::string e = exchange(sh, rs);
        if (!e.empty()) return e;
    }
    PT_TRY(pt_finish(rs.ctx));
    HIP_TRY(hipStreamSynchronize(rs.comm));
    PT_TRY(pt_get_stats(rs.ctx, &sh.stats[rs.rank]));
    if (rs.rank == 0) {
        sh.final_image.resize(sh.image_floats);
        HIP_TRY(hipMemcpy(sh.final_image.data(), tile ? rs.merged : rs.staging, sh.image_floats * sizeof(float), hipMemcpyDeviceToHost));
    }
    return "";
#undef PT_TRY
}

void rank_thread(Shared* sh, RankState* rs) {
    const std::string e = rank_body(*sh, *rs);
    if (!e.empty()) fail(*sh, rs->rank, e);
    // release what the rank holds, whatever happened
    (void)hipSetDevice(rs->device);
    if (rs->comm) (void)hipStreamSynchronize(rs->comm);
    if (rs->ctx) {
        Shader::SetContext(nullptr);
        pt_destroy(rs->ctx);
    }
    for (float* p : {rs->staging, rs->gathered, rs->merged})
        if (p) (void)hipFree(p);
    if (rs->copied) (void)hipEventDestroy(rs->copied);
    if (rs->merged_done) (void)hipEventDestroy(rs->merged_done);
    if (rs->render) (void)hipStreamDestroy(rs->render);
    if (rs->comm) (void)hipStreamDestroy(rs->comm);
}

}  // namespace

int RunRenderLoopMulti(const RenderOptions& options, const MultiGpuOptions& multi, Framebuffer& fb, RenderResult& result) {
    Shared sh;
    sh.opt = options;
    sh.multi = multi;
    sh.ranks = (int)multi.devices.size();
    if (sh.ranks < 1) {
        result.error = "no devices given";
        return 1;
    }
    if (options.program_kind == 2) {
        result.error = "the CPU-tracer program keeps running sums in tile order and is not partitioned: use the single-GPU loop";
        return 1;
    }
    if (multi.block_rows < 1) {
        result.error = "block_rows must be at least 1";
        return 1;
    }
    Workload w;
    if (!PrepareWorkload(sh.opt, w, &result.error)) return 1;
    sh.workload = &w;
    const int width = sh.opt.width, height = sh.opt.height;
    result.width = width;
    result.height = height;
    result.triangles = (int)(w.mesh.indices.size() / 3);
    sh.image_floats = (size_t)width * height * 4;
    sh.layout.width = width; sh.layout.height = height; sh.layout.block_rows = multi.block_rows; sh.layout.ranks = sh.ranks;
    const std::set<int> distinct(multi.devices.begin(), multi.devices.end());
    const bool own_devices = (int)distinct.size() == sh.ranks;
    sh.use_rccl = multi.exchange == MultiGpuOptions::EXCHANGE_RCCL || (multi.exchange == MultiGpuOptions::EXCHANGE_AUTO && own_devices);
    if (sh.use_rccl && !own_devices) {
        result.error = "RCCL needs a device of its own for every rank; ranks that share a GPU use the host exchange";
        return 1;
    }
    static Rccl rccl;   // opened once per process
    if (sh.use_rccl) {
        if (!rccl.lib && !rccl.open(&result.error)) return 1;
        sh.rccl = &rccl;
        sh.comms.assign((size_t)sh.ranks, nullptr);
        const ncclResult_t r = rccl.CommInitAll(sh.comms.data(), sh.ranks, multi.devices.data());
        if (r != ncclSuccess) {
            result.error = std::string("ncclCommInitAll: ") + rccl.GetErrorString(r);
            return 1;
        }
    } else {
        const bool tile = multi.partition == MultiGpuOptions::TILE;
        const size_t part = tile ? sh.layout.packed_floats() : sh.image_floats;
        sh.host_parts.assign((size_t)sh.ranks, std::vector<float>(part));
        sh.host_merged.resize(tile ? part * sh.ranks : part);
    }
    result.exchange = sh.use_rccl ? "rccl" : "host";
    Barrier barrier(sh.ranks);
    sh.barrier = &barrier;
    sh.stats.resize((size_t)sh.ranks);
    std::vector<RankState> ranks((size_t)sh.ranks);
    const auto start = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> threads;
        for (int r = 0; r < sh.ranks; ++r) {
            ranks[(size_t)r].rank = r;
            ranks[(size_t)r].device = multi.devices[(size_t)r];
            threads.emplace_back(rank_thread, &sh, &ranks[(size_t)r]);
        }
        for (std::thread& t : threads) t.join();
    }
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    if (sh.use_rccl && !sh.comms_aborted)
        for (ncclComm_t c : sh.comms)
            if (c) (void)(sh.failed ? rccl.CommAbort(c) : rccl.CommDestroy(c));
    if (sh.failed) {
        result.error = sh.error;
        return 1;
    }
    for (const PtStats& st : sh.stats) {
        result.rays += st.rays;
        result.rays_traced += st.rays_traced;
        result.samples += st.samples;
        result.device_ms = std::max(result.device_ms, st.total_dispatch_ms);
    }
    fb.create(width, height);
    if (multi.partition == MultiGpuOptions::SAMPLE) {
        // sums -> mean (Framebuffer::storeMean turns it back into the running sum the reference's CPU path keeps)
        const float scale = 1.0f / (float)sh.opt.passes;
        for (size_t i = 0; i < sh.final_image.size(); ++i) sh.final_image[i] *= scale;
    }
    fb.storeMean(sh.final_image.data(), sh.opt.passes);
    if (!sh.opt.quiet)
        printf("%d ranks (%s partition, %s exchange): %d samples per pixel in %.3f s, %.1f Mrays/s evaluated\n", sh.ranks,
               multi.partition == MultiGpuOptions::TILE ? "tile" : "sample", result.exchange.c_str(), sh.opt.passes, result.seconds,
               (double)result.rays_traced / result.seconds * 1e-6);
    if (!sh.opt.output_tga.empty() && !WriteTga(sh.opt.output_tga.c_str(), fb)) {
        result.error = "cannot write " + sh.opt.output_tga;
        return 1;
    }
    return 0;
}

}  // namespace pthost
