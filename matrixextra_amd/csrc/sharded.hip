// sharded.hip — CSR x dense SpMM row-sharded over the GPUs of one node with an RCCL all-gather of C, behind the C-ABI.
//
// BASELINE.json north_star: "Shard SpMM by row-blocks across the 8 GPUs of one node with an RCCL all-gather of C over xGMI",
// with the host code staying in R / C++.  The exports (mx_tcrossprod_csr_dense_*, src/matmul.cpp:316-375) return an R matrix
// in HOST memory, so their sharded form (mx_set_devices, api.hip) needs no collective: every device downloads its rows.
// This file is the form for a DEVICE-resident consumer — the vignette's L-BFGS loop keeps multiplying one matrix
// (Introducing_MatrixExtra.Rmd:452-470): the matrix is cut ONCE into one row block per device (balanced by entries + rows,
// or equal rows), each block stays on its device with AUTO's kept plan, every product runs on all devices at once and ONE
// in-place ncclAllGather of equal slots leaves the full row-major C on EVERY device.  One process, one host thread per
// device (kernel launches and uploads go out in parallel), one ncclCommInitAll communicator per device; the all-gather of
// product k runs on its own stream under product k + 1 (two gathered buffers alternate).
//
// RCCL is loaded at first use with dlopen("librccl.so.1") — libmxgpu.so itself links libamdhip64 only, and a process that
// already holds an RCCL (PyTorch ships its own copy under that SONAME) gets that same copy instead of a second one.
//
// Layout of a gathered buffer: ndev slots of slot_rows x n elements, row-major, ld = n; block r = rows [cuts[r], cuts[r+1])
// of the product sits at the head of slot r (the rows of a slot past its block are never written).  With equal_rows the
// slots are the blocks — row i of C is row i of the buffer, contiguous m x n.
#include "mx_common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" const char *mx_last_error(void);

namespace mx {
int xfer_h2d(void *dst_dev, const void *src_host, size_t bytes);
int xfer_d2h(void *dst_host, const void *src_dev, size_t bytes);
}

namespace {

// ---------------------------------------------------------------------------------------------------- RCCL, loaded late
struct Rccl {
    void *lib = nullptr;
    std::string why;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString; }
};
static Rccl &rccl()
{
    static Rccl r = [] {
        Rccl q;
        const char *env = getenv("MXGPU_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            q.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (q.lib) break;
            q.why = dlerror();
        }
        if (!q.lib) return q;
        auto sym = [&](const char *s) { return dlsym(q.lib, s); };
        q.CommInitAll = (decltype(q.CommInitAll))sym("ncclCommInitAll");
        q.CommDestroy = (decltype(q.CommDestroy))sym("ncclCommDestroy");
        q.AllGather = (decltype(q.AllGather))sym("ncclAllGather");
        q.GroupStart = (decltype(q.GroupStart))sym("ncclGroupStart");
        q.GroupEnd = (decltype(q.GroupEnd))sym("ncclGroupEnd");
        q.GetErrorString = (decltype(q.GetErrorString))sym("ncclGetErrorString");
        q.GetVersion = (decltype(q.GetVersion))sym("ncclGetVersion");
        if (!q.ok()) q.why = "librccl is missing one of ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd";
        return q;
    }();
    return r;
}
#define MX_NCCL(expr)                                                                                     \
    do {                                                                                                  \
        ncclResult_t _r = (expr);                                                                         \
        if (_r != ncclSuccess) return mx::set_error("%s failed: %s", #expr, rccl().GetErrorString(_r));   \
    } while (0)

// ---------------------------------------------------------------------------------------------------- one thread per device
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has = false, done = true, quit = false;
    int rc = 0;
    std::string err;
    void loop()
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return has || quit; });
            if (quit && !has) return;
            std::function<int()> j = std::move(job);
            has = false;
            lk.unlock();
            const int r = j();
            std::string e = r ? mx_last_error() : "";
            lk.lock();
            rc = r; err = std::move(e); done = true;
            cv.notify_all();
        }
    }
    void start(std::function<int()> j)
    {
        { std::lock_guard<std::mutex> lk(mu); job = std::move(j); has = true; done = false; }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// What one device holds (several shards of a test run may share it)
struct Dev {
    int dev = 0;
    hipStream_t cs = nullptr;                 // the collective's stream
    void *C[2] = {nullptr, nullptr};          // gathered buffers (alternate from product to product)
    size_t C_cap = 0;
    void *B = nullptr;                        // replicated dense operand (mx_spmm_sharded_run uploads it here)
    size_t B_cap = 0;
    void *T = nullptr;                        // device 0 only: column-major staging of the result for the host copy
    size_t T_cap = 0;
    hipEvent_t gathered[2] = {nullptr, nullptr};
    ncclComm_t comm = nullptr;
};
// One row block of the matrix with what AUTO keeps per matrix (matrixextra_amd/device.py DeviceCSR, restated)
struct Shard {
    Dev *d = nullptr;
    int r0 = 0, r1 = 0;
    int64_t nnz = 0;
    hipStream_t st = nullptr;
    hipEvent_t product = nullptr;             // the block's product has been queued up to here
    int32_t *p = nullptr, *j = nullptr;
    double *x = nullptr;
    float prof[MX_PROFILE_LEN];
    bool have_prof = false;
    mx_spmm_plan *plan = nullptr;
    bool plan_tried = false, plan_ready = false;
    int sorted = -1;
    char kernel[48] = "none";
};

__global__ __launch_bounds__(256)
void slots_to_colmajor_kernel(int rows, int n, int row0, size_t ld_out, const char *__restrict__ slot, char *__restrict__ out, int sz)
{
    // 32 x 32 tiles through LDS: reads along a row of the slot, writes along a column of the column-major result
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int r_base = blockIdx.x * 32, c_base = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r_base + k, c = c_base + tx;
        double v = 0.0;
        if (r < rows && c < n) {
            if (sz == 8) v = ((const double *)slot)[(size_t)r * n + c];
            else v = (double)((const float *)slot)[(size_t)r * n + c];
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c_base + k, r = r_base + tx;
        if (r < rows && c < n) {
            if (sz == 8) ((double *)out)[(size_t)c * ld_out + row0 + r] = tile[tx][k];
            else ((float *)out)[(size_t)c * ld_out + row0 + r] = (float)tile[tx][k];      // (a float widened and narrowed: the same bits)
        }
    }
}

}  // namespace

struct mx_spmm_sharded {
    int m = 0, K = 0, slot_rows = 0, flip = 0, last = -1, last_n = 0, last_dt = 0;
    bool uses_rccl = false, equal_rows = false;
    std::vector<int> cuts;
    std::vector<std::unique_ptr<Dev>> devs;
    std::vector<Shard> sh;
    std::vector<std::unique_ptr<Worker>> wk;
    std::mutex mu;                            // one call at a time
};

namespace {

static int grow(void **p, size_t *cap, size_t bytes)
{
    if (*cap >= bytes && *p) return 0;
    if (*p) { MX_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    MX_HIP(hipMalloc(p, bytes ? bytes : 16));
    *cap = bytes;
    return 0;
}

// every worker runs f(shard index) on its device; the first failure's message becomes the caller's
static int on_all(mx_spmm_sharded *h, const std::function<int(int)> &f)
{
    const int ns = (int)h->sh.size();
    for (int k = 0; k < ns; k++) {
        const int dev = h->sh[k].d->dev;
        h->wk[k]->start([=, &f]() -> int {
            if (hipSetDevice(dev) != hipSuccess) return mx::set_error("hipSetDevice(%d) failed", dev);
            return f(k);
        });
    }
    int rc = 0;
    std::string first;
    for (int k = 0; k < ns; k++)
        if (h->wk[k]->wait() && !rc) { rc = 1; first = "device " + std::to_string(h->sh[k].d->dev) + ": " + h->wk[k]->err; }
    return rc ? mx::set_error("%s", first.c_str()) : 0;
}

// AUTO with what the matrix keeps (profile, sortedness, the plan): device.py spmm(keep_plan=True), in C++
static int kept_spmm(Shard &s, int K, int n, int dt, const void *B, size_t ldb, void *C, size_t ldc)
{
    const int m = s.r1 - s.r0;
    if (m == 0) return 0;
    const size_t sz = dt == MX_F64 ? 8 : 4;
    if (s.nnz == 0) {                                   // reference early-out (matmul.cpp:128-129): all zeros
        MX_HIP(hipMemsetAsync(C, 0, (size_t)m * ldc * sz, s.st));
        return 0;
    }
    if (!s.have_prof) {
        void *ws = nullptr;
        MX_HIP(hipMalloc(&ws, mxd_csr_profile_workspace_bytes(K) + 64));
        const int rc = mxd_csr_profile(m, K, s.nnz, s.p, s.j, s.prof, ws, s.st);
        (void)hipFree(ws);
        if (rc) return 1;
        s.have_prof = true;
    }
    int pick = 0;
    if (mxd_spmm_auto_algo3(m, n, K, s.nnz, 1, dt, B, ldb, C, ldc, 0, s.prof, &pick)) return 1;
    if (pick == MX_SPMM_PLANNED) {
        if (!s.plan_tried) {
            int ready = 0;
            if (mxd_spmm_plan_create_auto(m, K, s.p, s.j, s.x, 0, s.st, &s.plan, &ready)) return 1;
            s.plan_tried = true; s.plan_ready = ready != 0;
        }
        bool use = s.plan_ready;
        if (use) {
            double imb = 0.0;
            if (mxd_spmm_plan_imbalance(s.plan, n, dt, &imb)) return 1;
            use = imb <= mxd_spmm_plan_imbalance_limit(dt);
        }
        if (use) {
            snprintf(s.kernel, sizeof(s.kernel), "spmm_plan_kernel");
            return mxd_spmm_plan_run(s.plan, n, B, ldb, C, ldc, dt, 0, 0, -1, s.st);
        }
        pick = MX_SPMM_ROWSPLIT;                        // the plan would pad or tail too much: AUTO's fallback
    }
    if ((pick == MX_SPMM_SLAB || pick == MX_SPMM_TILE) && s.sorted < 0) {
        int32_t *flag = nullptr;
        MX_HIP(hipMalloc((void **)&flag, 16));
        int sorted = 0;
        const int rc = mxd_csr_rows_sorted(m, s.p, s.j, flag, &sorted, s.st);
        (void)hipFree(flag);
        if (rc) return 1;
        s.sorted = sorted;
    }
    if (mxd_spmm_csr_dense_ex3(m, n, K, s.nnz, s.p, s.j, s.x, B, ldb, C, ldc, dt, 0, pick, s.sorted > 0 ? 1 : 0, 0, 0, s.prof, s.st)) return 1;
    snprintf(s.kernel, sizeof(s.kernel), "%s", mxd_spmm_last_kernel());
    return 0;
}

static void free_shard(Shard &s)
{
    if (s.plan) (void)mxd_spmm_plan_destroy(s.plan);
    if (s.p) (void)hipFree(s.p);
    if (s.j) (void)hipFree(s.j);
    if (s.x) (void)hipFree(s.x);
    if (s.product) (void)hipEventDestroy(s.product);
    if (s.st) (void)hipStreamDestroy(s.st);
    s = Shard();
}

// the product of every block into gathered buffer `buf`, then the all-gather; B_of(k): the operand on shard k's device
static int run_products(mx_spmm_sharded *h, int n, int dt, const std::function<const void *(int)> &B_of, size_t ldb)
{
    const size_t sz = dt == MX_F64 ? 8 : 4, slot_bytes = (size_t)h->slot_rows * n * sz;
    const int buf = h->flip, ns = (int)h->sh.size(), K = h->K;
    const int nslots = ns;
    if (on_all(h, [&](int k) -> int {
            Shard &s = h->sh[k];
            Dev &d = *s.d;
            if (&h->sh[k] == &h->sh[0] || h->sh[k - 1].d != s.d) {         // the device's first shard sizes its buffers
                for (int b = 0; b < 2; b++) {
                    size_t cap = d.C_cap;
                    if (grow(&d.C[b], &cap, slot_bytes * nslots)) return 1;
                    if (b == 1) d.C_cap = cap;
                }
            }
            return 0;
        })) return 1;
    if (on_all(h, [&](int k) -> int {
            Shard &s = h->sh[k];
            Dev &d = *s.d;
            // this buffer was last read by the all-gather two products ago
            if (d.gathered[buf]) MX_HIP(hipStreamWaitEvent(s.st, d.gathered[buf], 0));
            char *slot = (char *)d.C[buf] + slot_bytes * (size_t)k;
            if (kept_spmm(s, K, n, dt, B_of(k), ldb, slot, (size_t)n)) return 1;
            MX_HIP(hipEventRecord(s.product, s.st));
            return 0;
        })) return 1;
    // the exchange: one in-place all-gather of equal slots per device, queued by this thread as ONE group
    for (auto &dp : h->devs) {
        MX_HIP(hipSetDevice(dp->dev));
        for (auto &s : h->sh) if (s.d == dp.get()) MX_HIP(hipStreamWaitEvent(dp->cs, s.product, 0));
    }
    if (h->uses_rccl) {
        Rccl &R = rccl();
        MX_NCCL(R.GroupStart());
        for (size_t r = 0; r < h->devs.size(); r++) {
            Dev &d = *h->devs[r];
            ncclResult_t q = R.AllGather((char *)d.C[buf] + slot_bytes * r, d.C[buf], slot_bytes, ncclInt8, d.comm, d.cs);
            if (q != ncclSuccess) { (void)R.GroupEnd(); return mx::set_error("ncclAllGather failed: %s", R.GetErrorString(q)); }
        }
        MX_NCCL(R.GroupEnd());
    }
    for (auto &dp : h->devs) {
        MX_HIP(hipSetDevice(dp->dev));
        if (!dp->gathered[buf]) MX_HIP(hipEventCreateWithFlags(&dp->gathered[buf], hipEventDisableTiming));
        MX_HIP(hipEventRecord(dp->gathered[buf], dp->cs));
    }
    h->last = buf; h->last_n = n; h->last_dt = dt;
    h->flip ^= 1;
    return 0;
}

static int sync_all(mx_spmm_sharded *h)
{
    int cur = 0;
    (void)hipGetDevice(&cur);
    int rc = 0;
    for (auto &dp : h->devs) {
        if (hipSetDevice(dp->dev) != hipSuccess || hipStreamSynchronize(dp->cs) != hipSuccess) rc = 1;
        for (auto &s : h->sh) if (s.d == dp.get() && hipStreamSynchronize(s.st) != hipSuccess) rc = 1;
    }
    (void)hipSetDevice(cur);
    return rc ? mx::set_error("mx_spmm_sharded: a device failed while synchronising (%s)", hipGetErrorString(hipGetLastError())) : 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- the C-ABI
// host arithmetic only (tests/test_sharded_layout.py): the row cuts and the slot size for a device list
extern "C" int mx_spmm_sharded_layout(const int32_t *indptr, int m, int ndev, int dense_cols, int dense_bytes, int equal_rows,
                                      int *cuts, int *slot_rows)
{
    MX_REQUIRE(indptr && cuts && slot_rows && m >= 0 && ndev >= 1, "mx_spmm_sharded_layout: bad arguments");
    if (equal_rows) {
        const int per = (int)mx::ceil_div(m, ndev);
        for (int r = 0; r <= ndev; r++) cuts[r] = (int)std::min<int64_t>(m, (int64_t)per * r);
    } else {
        // balanced by what a block COSTS ON ITS DEVICE: one row of B read per entry, one row of C written per row — the same
        // n * s bytes either way, so cost(r rows) = entries + rows.  (mx_partition_rows balances the sharded EXPORTS: 12 bytes
        // up per entry against a result row down over PCIe — another ratio.)
        (void)dense_cols; (void)dense_bytes;
        const double base = (double)indptr[0], total = (double)indptr[m] - base + (double)m;
        cuts[0] = 0;
        for (int k = 1; k < ndev; k++) {
            const double target = total * (double)k / (double)ndev;
            int lo = cuts[k - 1], hi = m;                           // first r with cost(r) >= target
            while (lo < hi) {
                const int mid = lo + (hi - lo) / 2;
                if ((double)indptr[mid] - base + (double)mid < target) lo = mid + 1; else hi = mid;
            }
            cuts[k] = lo;
        }
        cuts[ndev] = m;
    }
    int s = 0;
    for (int r = 0; r < ndev; r++) s = std::max(s, cuts[r + 1] - cuts[r]);
    // whole 256-byte lines per slot whatever n and the element size are (slot_rows a multiple of 64)
    *slot_rows = (int)std::min<int64_t>(INT_MAX - 63, ((int64_t)s + 63) / 64 * 64);
    if (equal_rows) *slot_rows = (int)mx::ceil_div(m, ndev);          // (the slots ARE the blocks: C stays contiguous)
    return 0;
}

extern "C" int mx_spmm_sharded_create(const int *devices, int ndev, int m, int K, const int32_t *indptr, const int32_t *indices,
                                      const double *values, int equal_rows, mx_spmm_sharded **out)
{
    MX_REQUIRE(out, "mx_spmm_sharded_create: null result pointer");
    *out = nullptr;
    MX_REQUIRE(devices && ndev >= 1 && ndev <= 64, "mx_spmm_sharded_create: 1 .. 64 devices");
    MX_REQUIRE(m >= 0 && K >= 0 && indptr, "mx_spmm_sharded_create: bad matrix arguments");
    MX_REQUIRE(indptr[m] == 0 || (indices && values), "mx_spmm_sharded_create: null indices / values");
    int count = 0;
    MX_HIP(hipGetDeviceCount(&count));
    std::vector<int> distinct;
    for (int k = 0; k < ndev; k++) {
        MX_REQUIRE(devices[k] >= 0 && devices[k] < count, "mx_spmm_sharded_create: device %d of %d", devices[k], count);
        if (std::find(distinct.begin(), distinct.end(), devices[k]) == distinct.end()) distinct.push_back(devices[k]);
    }
    // distinct devices: one shard each, RCCL between them.  ONE device listed several times: its shards share the device's
    // gathered buffer and nothing is exchanged (how a one-GPU box runs the slot arithmetic of N shards).
    MX_REQUIRE((int)distinct.size() == ndev || distinct.size() == 1,
               "mx_spmm_sharded_create: list every device once (or one device several times: shards without an exchange)");
    int cur = 0;
    MX_HIP(hipGetDevice(&cur));
    std::unique_ptr<mx_spmm_sharded> h(new mx_spmm_sharded());
    h->m = m; h->K = K; h->equal_rows = equal_rows != 0;
    h->cuts.resize((size_t)ndev + 1);
    // (the balance is set for a typical B of 128 f64 columns; what matters is nnz against rows, not the exact n)
    if (mx_spmm_sharded_layout(indptr, m, ndev, 128, 8, equal_rows, h->cuts.data(), &h->slot_rows)) return 1;
    for (int dv : distinct) { h->devs.emplace_back(new Dev()); h->devs.back()->dev = dv; }
    h->sh.resize((size_t)ndev);
    for (int k = 0; k < ndev; k++) {
        Shard &s = h->sh[k];
        s.d = h->devs[distinct.size() == 1 ? 0 : (size_t)k].get();
        s.r0 = h->cuts[k]; s.r1 = h->cuts[k + 1];
        s.nnz = (int64_t)indptr[s.r1] - indptr[s.r0];
        h->wk.emplace_back(new Worker());
        Worker *w = h->wk.back().get();
        w->th = std::thread([w] { w->loop(); });
    }
    auto fail = [&](int) { mx_spmm_sharded *raw = h.release(); std::string e = mx_last_error(); (void)mx_spmm_sharded_destroy(raw); (void)hipSetDevice(cur); return mx::set_error("%s", e.c_str()); };
    // every block goes up on its own thread: rebased row pointers, its slice of the indices and values
    if (on_all(h.get(), [&](int k) -> int {
            Shard &s = h->sh[k];
            const int mb = s.r1 - s.r0;
            MX_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
            MX_HIP(hipEventCreateWithFlags(&s.product, hipEventDisableTiming));
            std::vector<int32_t> pl((size_t)mb + 1);
            for (int r = 0; r <= mb; r++) pl[r] = indptr[s.r0 + r] - indptr[s.r0];
            MX_HIP(hipMalloc((void **)&s.p, sizeof(int32_t) * ((size_t)mb + 1)));
            MX_HIP(hipMalloc((void **)&s.j, sizeof(int32_t) * (size_t)std::max<int64_t>(s.nnz, 4)));
            MX_HIP(hipMalloc((void **)&s.x, sizeof(double) * (size_t)std::max<int64_t>(s.nnz, 2)));
            if (mx::xfer_h2d(s.p, pl.data(), sizeof(int32_t) * ((size_t)mb + 1))) return 1;
            if (s.nnz && mx::xfer_h2d(s.j, indices + indptr[s.r0], sizeof(int32_t) * (size_t)s.nnz)) return 1;
            if (s.nnz && mx::xfer_h2d(s.x, values + indptr[s.r0], sizeof(double) * (size_t)s.nnz)) return 1;
            return 0;
        })) return fail(0);
    for (auto &dp : h->devs) {
        if (hipSetDevice(dp->dev) != hipSuccess || hipStreamCreateWithFlags(&dp->cs, hipStreamNonBlocking) != hipSuccess) {
            mx::set_error("mx_spmm_sharded_create: cannot create a stream on device %d", dp->dev);
            return fail(0);
        }
    }
    if ((int)distinct.size() == ndev) {
        Rccl &R = rccl();
        if (!R.ok()) { mx::set_error("mx_spmm_sharded_create: RCCL cannot be loaded (%s)", R.why.c_str()); return fail(0); }
        std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
        const ncclResult_t q = R.CommInitAll(comms.data(), ndev, distinct.data());
        if (q != ncclSuccess) { mx::set_error("ncclCommInitAll over %d device(s) failed: %s", ndev, R.GetErrorString(q)); return fail(0); }
        for (int k = 0; k < ndev; k++) h->devs[k]->comm = comms[k];
        h->uses_rccl = true;
    }
    (void)hipSetDevice(cur);
    *out = h.release();
    return 0;
}

extern "C" int mx_spmm_sharded_info(const mx_spmm_sharded *h, int *nshards, int *slot_rows, int *cuts, int *uses_rccl, int *rccl_version)
{
    MX_REQUIRE(h, "mx_spmm_sharded_info: null handle");
    if (nshards) *nshards = (int)h->sh.size();
    if (slot_rows) *slot_rows = h->slot_rows;
    if (cuts) std::copy(h->cuts.begin(), h->cuts.end(), cuts);
    if (uses_rccl) *uses_rccl = h->uses_rccl ? 1 : 0;
    if (rccl_version) { *rccl_version = 0; if (h->uses_rccl && rccl().GetVersion) (void)rccl().GetVersion(rccl_version); }
    return 0;
}

// B already on every device: B_dev[k] = the K x n row-major operand on shard k's device.  flags bit 0: return once everything
// is queued (mx_spmm_sharded_sync waits) — the all-gather of this product then runs under the next one.
extern "C" int mx_spmm_sharded_run_dev(mx_spmm_sharded *h, int n, int dense_dtype, const void *const *B_dev, size_t ldb, int flags)
{
    MX_REQUIRE(h && B_dev, "mx_spmm_sharded_run_dev: null argument");
    MX_REQUIRE(n >= 1 && ldb >= (size_t)n, "mx_spmm_sharded_run_dev: bad n / ldb");
    MX_REQUIRE(dense_dtype == MX_F64 || dense_dtype == MX_F32, "mx_spmm_sharded_run_dev: unsupported dense dtype %d", dense_dtype);
    for (size_t k = 0; k < h->sh.size(); k++)
        MX_REQUIRE(B_dev[k] || h->K == 0, "mx_spmm_sharded_run_dev: no operand for shard %zu (B_dev[%zu] is NULL)", k, k);
    std::lock_guard<std::mutex> lk(h->mu);
    int cur = 0;
    MX_HIP(hipGetDevice(&cur));
    const int rc = run_products(h, n, dense_dtype, [&](int k) { return B_dev[k]; }, ldb) || (!(flags & 1) && sync_all(h));
    (void)hipSetDevice(cur);
    return rc;
}

// B on the host (K x n row-major = R's column-major n x K matrix Y of tcrossprod_csr_dense), replicated to every device;
// C_host (optional): the result from the first device — row-major m x n (ldc >= n) or column-major (ldc >= m).
extern "C" int mx_spmm_sharded_run(mx_spmm_sharded *h, int n, int dense_dtype, const void *B_host, size_t ldb, void *C_host, size_t ldc,
                                   int colmajor_out)
{
    MX_REQUIRE(h && B_host, "mx_spmm_sharded_run: null argument");
    MX_REQUIRE(n >= 1 && ldb >= (size_t)n, "mx_spmm_sharded_run: bad n / ldb");
    MX_REQUIRE(dense_dtype == MX_F64 || dense_dtype == MX_F32, "mx_spmm_sharded_run: unsupported dense dtype %d", dense_dtype);
    MX_REQUIRE(!C_host || ldc >= (size_t)(colmajor_out ? h->m : n), "mx_spmm_sharded_run: ldc too small");
    std::lock_guard<std::mutex> lk(h->mu);
    int cur = 0;
    MX_HIP(hipGetDevice(&cur));
    const size_t sz = dense_dtype == MX_F64 ? 8 : 4, b_bytes = (size_t)h->K * ldb * sz;
    struct Back { int d; ~Back() { (void)hipSetDevice(d); } } back{cur};
    // one upload per distinct device, by the device's first shard (its thread; the uploads run side by side)
    if (on_all(h, [&](int k) -> int {
            Shard &s = h->sh[k];
            if (k > 0 && h->sh[k - 1].d == s.d) return 0;
            if (grow(&s.d->B, &s.d->B_cap, b_bytes)) return 1;
            return b_bytes ? mx::xfer_h2d(s.d->B, B_host, b_bytes) : 0;
        })) return 1;
    if (run_products(h, n, dense_dtype, [&](int k) { return (const void *)h->sh[k].d->B; }, ldb)) return 1;
    if (sync_all(h)) return 1;
    if (!C_host || h->m == 0) return 0;
    Dev &d = *h->devs[0];
    MX_HIP(hipSetDevice(d.dev));
    const size_t slot_bytes = (size_t)h->slot_rows * n * sz;
    const char *G = (const char *)d.C[h->last];
    if (!colmajor_out) {
        for (size_t r = 0; r + 1 < h->cuts.size(); r++) {
            const int rows = h->cuts[r + 1] - h->cuts[r];
            if (!rows) continue;
            char *dst = (char *)C_host + (size_t)h->cuts[r] * ldc * sz;
            if (ldc == (size_t)n) { if (mx::xfer_d2h(dst, G + slot_bytes * r, (size_t)rows * n * sz)) return 1; }
            else MX_HIP(hipMemcpy2D(dst, ldc * sz, G + slot_bytes * r, (size_t)n * sz, (size_t)n * sz, (size_t)rows, hipMemcpyDeviceToHost));
        }
        return 0;
    }
    // column-major for R: transposed on the device into one m x n column-major block, one download
    if (grow(&d.T, &d.T_cap, (size_t)h->m * n * sz)) return 1;
    for (size_t r = 0; r + 1 < h->cuts.size(); r++) {
        const int rows = h->cuts[r + 1] - h->cuts[r];
        if (!rows) continue;
        hipLaunchKernelGGL(slots_to_colmajor_kernel, dim3((unsigned)mx::ceil_div(rows, 32), (unsigned)mx::ceil_div(n, 32)), dim3(256), 0, d.cs,
                           rows, n, h->cuts[r], (size_t)h->m, G + slot_bytes * r, (char *)d.T, (int)sz);
    }
    MX_LAUNCH_CHECK();
    MX_HIP(hipStreamSynchronize(d.cs));
    if (ldc == (size_t)h->m) return mx::xfer_d2h(C_host, d.T, (size_t)h->m * n * sz);
    MX_HIP(hipMemcpy2D(C_host, ldc * sz, d.T, (size_t)h->m * sz, (size_t)h->m * sz, (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mx_spmm_sharded_sync(mx_spmm_sharded *h)
{
    MX_REQUIRE(h, "mx_spmm_sharded_sync: null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    return sync_all(h);
}

// the last product's gathered result on shard k's device: slot r (rows cuts[r] .. cuts[r+1] of C) starts at row r * slot_rows of
// a row-major buffer with ld = n.  Valid until the product after next (two buffers alternate).
extern "C" int mx_spmm_sharded_result(const mx_spmm_sharded *h, int k, void **C_dev, int *n, int *dense_dtype, int *device)
{
    MX_REQUIRE(h && C_dev, "mx_spmm_sharded_result: null argument");
    MX_REQUIRE(k >= 0 && k < (int)h->sh.size(), "mx_spmm_sharded_result: shard %d of %d", k, (int)h->sh.size());
    MX_REQUIRE(h->last >= 0, "mx_spmm_sharded_result: no product has run yet");
    *C_dev = h->sh[k].d->C[h->last];
    if (n) *n = h->last_n;
    if (dense_dtype) *dense_dtype = h->last_dt;
    if (device) *device = h->sh[k].d->dev;
    return 0;
}

// name of the kernel shard k's last product ran (reporting only)
extern "C" const char *mx_spmm_sharded_kernel(const mx_spmm_sharded *h, int k)
{
    return h && k >= 0 && k < (int)h->sh.size() ? h->sh[k].kernel : "none";
}

extern "C" int mx_spmm_sharded_destroy(mx_spmm_sharded *h)
{
    if (!h) return 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)sync_all(h);
    // per-thread workspaces of the workers (AUTO's scratch) go back on the thread that owns them
    for (size_t k = 0; k < h->wk.size(); k++) {
        Shard *s = &h->sh[k];
        const int dev = s->d ? s->d->dev : 0;
        h->wk[k]->start([s, dev]() -> int { (void)hipSetDevice(dev); free_shard(*s); (void)mxd_release_workspaces(); return 0; });
    }
    for (auto &w : h->wk) { (void)w->wait(); w->stop(); }
    for (auto &dp : h->devs) {
        (void)hipSetDevice(dp->dev);
        if (dp->comm && rccl().ok()) (void)rccl().CommDestroy(dp->comm);
        for (int b = 0; b < 2; b++) { if (dp->C[b]) (void)hipFree(dp->C[b]); if (dp->gathered[b]) (void)hipEventDestroy(dp->gathered[b]); }
        if (dp->B) (void)hipFree(dp->B);
        if (dp->T) (void)hipFree(dp->T);
        if (dp->cs) (void)hipStreamDestroy(dp->cs);
    }
    (void)hipSetDevice(cur);
    delete h;
    return 0;
}
