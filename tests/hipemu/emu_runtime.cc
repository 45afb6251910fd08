// hipemu runtime: fibers for the lanes of a workgroup, host memory for HBM.  TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <chrono>
#include <mutex>
#include <vector>

namespace hipemu {

Idx threadIdx_, blockIdx_, blockDim_, gridDim_;
int lane_ = 0;

namespace {

enum State { RUNNABLE, AT_BARRIER, AT_WAVE, DONE };
struct Fiber {
    void *sp;
    State state;
    const void *site;
    unsigned long long posted;
    Idx tid;
    int flat;
};

constexpr size_t STACK_BYTES = 512 << 10;
constexpr int MAX_THREADS = 1024;
constexpr size_t MAX_LDS = 160 << 10;

char *g_stacks = nullptr;
Fiber g_fibers[MAX_THREADS];
void *g_sched_sp = nullptr;
int g_cur = -1;
const Body *g_body = nullptr;
alignas(64) unsigned char g_lds[MAX_LDS];
const void *g_kernarg = nullptr;
unsigned long long g_snap[MAX_THREADS / 64][64];
unsigned long long g_snap_mask[MAX_THREADS / 64];
std::recursive_mutex g_lock;
unsigned long long g_rng = 0;
bool g_scramble = false;

// callee-saved registers + stack pointer (System V x86-64)
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

void yield_to_scheduler(State s) {
    Fiber &f = g_fibers[g_cur];
    f.state = s;
    hipemu_switch(&f.sp, g_sched_sp);
}

extern "C" void hipemu_fiber_main() {
    (*g_body)();
    yield_to_scheduler(DONE);
    abort();   // a finished fiber is never resumed
}

void prepare(Fiber &f, int k) {
    // stack top: [r15 r14 r13 r12 rbx rbp | return address = hipemu_fiber_main | padding]; rsp is 16-byte aligned + 8 at function entry
    char *top = g_stacks + (size_t)(k + 1) * STACK_BYTES;
    void **sp = (void **)(top - 64);
    sp[6] = (void *)&hipemu_fiber_main;
    for (int q = 0; q < 6; q++) sp[q] = nullptr;
    f.sp = sp;
    f.state = RUNNABLE;
}

unsigned next_rand() {
    g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
    return (unsigned)(g_rng >> 33);
}

void run_block(int nthreads) {
    std::vector<int> order(nthreads);
    for (int k = 0; k < nthreads; k++) order[k] = k;
    const int nwaves = (nthreads + 63) / 64;
    int done = 0;
    while (done < nthreads) {
        if (g_scramble)
            for (int k = nthreads - 1; k > 0; k--) std::swap(order[k], order[next_rand() % (unsigned)(k + 1)]);
        bool ran = false;
        for (int q = 0; q < nthreads; q++) {
            const int k = order[q];
            Fiber &f = g_fibers[k];
            if (f.state != RUNNABLE) continue;
            ran = true;
            g_cur = k;
            threadIdx_ = f.tid;
            lane_ = f.flat & 63;
            hipemu_switch(&g_sched_sp, f.sp);
            if (f.state == DONE) done++;
        }
        if (ran) continue;
        // nobody can run: complete wave operations first (one call site per wave at a time), then the workgroup barrier
        bool released = false;
        for (int w = 0; w < nwaves; w++) {
            const void *site = nullptr;
            for (int l = 0; l < 64 && w * 64 + l < nthreads; l++) {
                Fiber &f = g_fibers[w * 64 + l];
                if (f.state == AT_WAVE && (site == nullptr || f.site < site)) site = f.site;
            }
            if (!site) continue;
            g_snap_mask[w] = 0;
            for (int l = 0; l < 64 && w * 64 + l < nthreads; l++) {
                Fiber &f = g_fibers[w * 64 + l];
                if (f.state == AT_WAVE && f.site == site) {
                    g_snap[w][l] = f.posted;
                    g_snap_mask[w] |= 1ull << l;
                    f.state = RUNNABLE;
                    released = true;
                }
            }
        }
        if (released) continue;
        int waiting = 0;
        for (int k = 0; k < nthreads; k++) waiting += g_fibers[k].state == AT_BARRIER;
        if (waiting + done != nthreads || waiting == 0) {
            fprintf(stderr, "hipemu: workgroup is stuck (%d at the barrier, %d done, %d threads)\n", waiting, done, nthreads);
            abort();
        }
        for (int k = 0; k < nthreads; k++)
            if (g_fibers[k].state == AT_BARRIER) g_fibers[k].state = RUNNABLE;
    }
}

}  // namespace

void sync_threads() { yield_to_scheduler(AT_BARRIER); }

void wave_meet(const void *site, unsigned long long v) {
    Fiber &f = g_fibers[g_cur];
    f.site = site;
    f.posted = v;
    yield_to_scheduler(AT_WAVE);
}
unsigned long long wave_posted(int lane) { return g_snap[g_fibers[g_cur].flat >> 6][lane]; }
unsigned long long wave_mask() { return g_snap_mask[g_fibers[g_cur].flat >> 6]; }
void *dynamic_lds() { return g_lds; }
const void *kernarg() { return g_kernarg; }

void launch(dim3 grid, dim3 block, size_t lds, const Body &body, const void *karg, size_t) {
    std::lock_guard<std::recursive_mutex> guard(g_lock);
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > MAX_THREADS || lds > MAX_LDS) {
        fprintf(stderr, "hipemu: launch with %d threads, %zu bytes of dynamic LDS\n", nthreads, lds);
        abort();
    }
    if (!g_stacks) {
        g_stacks = (char *)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == MAP_FAILED) { perror("hipemu: mmap"); abort(); }
        if (const char *v = getenv("HIPEMU_SCRAMBLE")) { g_scramble = true; g_rng = strtoull(v, nullptr, 10) * 2654435761ull + 1; }
    }
    g_body = &body;
    g_kernarg = karg;
    blockDim_ = Idx{block.x, block.y, block.z};
    gridDim_ = Idx{grid.x, grid.y, grid.z};
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    const size_t start = g_scramble ? next_rand() % nblocks : 0;
    const size_t stride = 1;   // (blocks in rotated order when scrambling: enough to break "block 0 first" assumptions)
    for (size_t n = 0; n < nblocks; n++) {
        const size_t b = (start + n * stride) % nblocks;
        blockIdx_ = Idx{(unsigned)(b % grid.x), (unsigned)(b / grid.x % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y))};
        memset(g_lds, 0xA5, lds);    // dynamic LDS starts as garbage, as on the device
        for (int k = 0; k < nthreads; k++) {
            Fiber &f = g_fibers[k];
            f.flat = k;
            f.tid = Idx{k % block.x, k / block.x % block.y, k / (block.x * block.y)};
            prepare(f, k);
        }
        run_block(nthreads);
    }
    g_body = nullptr;
}

}  // namespace hipemu

// ------------------------------------------------------------------------------------------------ runtime API
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) {
    *v = a == hipDeviceAttributeMaxSharedMemoryPerBlock ? (64 << 10) : 256;
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) {
    if (posix_memalign(p, 256, n ? n : 1)) return hipErrorInvalidValue;
    memset(*p, 0xCD, n);      // fresh device memory is garbage
    return hipSuccess;
}
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return posix_memalign(p, 256, n ? n : 1) ? hipErrorInvalidValue : hipSuccess; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemuEvent{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipemuEvent{0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes *a, const void *) {
    memset(a, 0, sizeof(*a));
    a->maxThreadsPerBlock = 1024;
    return hipSuccess;
}
