// TEST INFRASTRUCTURE ONLY — runtime of the host-side HIP stand-in (see hip/hip_runtime.h here).
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <sys/mman.h>
#include <time.h>
#include <ucontext.h>
#include <unistd.h>

#include <map>
#include <stdexcept>

namespace hipemu {

uint3_emu g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;

// Context switch between fibers.  glibc's swapcontext saves and restores the signal mask with a system call on every switch
// (two per yield, 64 yields per emulated wave operation: a third of the CPU suite's time went into rt_sigprocmask); on x86-64 the
// fibers switch with a dozen instructions instead - callee-saved registers and the stack pointer (System V ABI; the floating-point
// control state is the same in every fiber).  Other hosts keep ucontext.
#if defined(__x86_64__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
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
    .size hipemu_switch, .-hipemu_switch
)");
struct Context {
  void* sp = nullptr;
};
static void switch_context(Context* from, Context* to) { hipemu_switch(&from->sp, to->sp); }
static void make_context(Context* c, char* stack, size_t size, void (*entry)()) {
  // the frame hipemu_switch pops: six callee-saved registers, then `ret` into entry with the stack as a call would leave it
  // (return-address slot at a 16-byte boundary minus 8)
  void** sp = reinterpret_cast<void**>(reinterpret_cast<uintptr_t>(stack + size) & ~uintptr_t(15));
  *--sp = nullptr;                              // the return address entry() would see: it never returns
  *--sp = reinterpret_cast<void*>(entry);
  for (int i = 0; i < 6; ++i) *--sp = nullptr;
  c->sp = sp;
}
#else
#define HIPEMU_FAST_SWITCH 0
struct Context {
  ucontext_t uc;
};
static void switch_context(Context* from, Context* to) { swapcontext(&from->uc, &to->uc); }
static void make_context(Context* c, char* stack, size_t size, void (*entry)()) {
  getcontext(&c->uc);
  c->uc.uc_stack.ss_sp = stack;
  c->uc.uc_stack.ss_size = size;
  c->uc.uc_link = nullptr;
  makecontext(&c->uc, entry, 0);
}
#endif

namespace {
enum State { RUN, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Fiber {
  Context ctx;
  char* stack = nullptr;
  State st = RUN;
  uint3_emu tid;
};
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
Context sched_ctx;
int cur = -1;
const std::function<void()>* cur_body = nullptr;
struct WaveBuf {
  unsigned v[2][64];
  float a[2][64], b[2][64];
  unsigned long long q[2][128];
  int gen = 0;
};
std::vector<WaveBuf> waves;
std::vector<char*> stack_pool;

void fiber_main() {
  (*cur_body)();
  fibers[cur].st = DONE;
  switch_context(&fibers[cur].ctx, &sched_ctx);
  abort();   // a finished fiber is never resumed
}
void yield(State s) {
  fibers[cur].st = s;
  switch_context(&fibers[cur].ctx, &sched_ctx);
}
}  // namespace

static int g_tpb();
int lane_id() { return (cur % g_tpb()) & 63; }
void* dyn_shared();

static int wave_of_cur();
void sync_block() { yield(WAIT_BLOCK); }

unsigned wave_exchange(unsigned v, int src_lane) {
  WaveBuf& w = waves[wave_of_cur()];
  int g = w.gen & 1;
  w.v[g][lane_id()] = v;
  yield(WAIT_WAVE);
  // after release the scheduler bumped w.gen; our data sits in buffer g
  return w.v[g][src_lane & 63];
}

void wave_exchange2(float a, float b, const float** A, const float** B) {
  WaveBuf& w = waves[wave_of_cur()];
  int g = w.gen & 1;
  w.a[g][lane_id()] = a;
  w.b[g][lane_id()] = b;
  yield(WAIT_WAVE);
  *A = w.a[g];
  *B = w.b[g];
}

const unsigned long long* wave_publish2(unsigned long long a, unsigned long long b) {
  WaveBuf& w = waves[wave_of_cur()];
  int g = w.gen & 1;
  w.q[g][lane_id() * 2] = a;
  w.q[g][lane_id() * 2 + 1] = b;
  yield(WAIT_WAVE);
  return w.q[g];
}

static int sched_order() {
  const char* e = getenv("HIPEMU_ORDER");
  return e ? atoi(e) : 0;
}

// Runs `nblk` blocks (block indices ids[0..nblk)) to completion with all their fibers alive: one block at a time for an
// ordinary launch, the whole grid for a resident one.  Fiber i belongs to block i / n.
static std::vector<uint3_emu> g_fiber_block;       // block index per resident block slot
static std::vector<std::vector<float>> g_dyn_pool; // dynamic LDS per resident block slot
static int g_threads_per_block = 0;

static void run_blocks(dim3 block, const std::vector<uint3_emu>& ids, size_t shmem, const std::function<void()>& body) {
  const int n = block.x * block.y * block.z;
  const int nblk = (int)ids.size();
  const int nw = (n + 63) / 64;
  const int total = n * nblk;
  g_threads_per_block = n;
  g_fiber_block = ids;
  g_dyn_pool.resize(nblk);
  for (auto& d : g_dyn_pool) d.assign(shmem / sizeof(float) + 4, NAN);   // poisoned: reads of unwritten LDS show up
  fibers.resize(total);
  waves.assign((size_t)nw * nblk, WaveBuf());
  while ((int)stack_pool.size() < total) stack_pool.push_back((char*)malloc(kStack));
  cur_body = &body;
  for (int i = 0; i < total; ++i) {
    Fiber& f = fibers[i];
    const int t = i % n;
    f.st = RUN;
    f.stack = stack_pool[i];
    f.tid.x = t % block.x;
    f.tid.y = (t / block.x) % block.y;
    f.tid.z = t / (block.x * block.y);
    make_context(&f.ctx, f.stack, kStack, fiber_main);
  }
  const int order = sched_order();
  std::vector<int> seq(n);
  for (int i = 0; i < n; ++i) {
    if (order == 1) seq[i] = n - 1 - i;
    else if (order == 2) seq[i] = ((nw - 1 - (i >> 6)) << 6 | (i & 63)) < n ? ((nw - 1 - (i >> 6)) << 6 | (i & 63)) : i;
    else seq[i] = i;
  }
  for (;;) {
    bool progressed = false;
    int done = 0;
    for (int b = 0; b < nblk; ++b)
      for (int k = 0; k < n; ++k) {
        const int i = b * n + seq[k];
        if (fibers[i].st == DONE) { ++done; continue; }
        if (fibers[i].st != RUN) continue;
        cur = i;
        g_threadIdx = fibers[i].tid;
        g_blockIdx = ids[b];
        switch_context(&sched_ctx, &fibers[i].ctx);
        progressed = true;
        if (fibers[i].st == DONE) ++done;
      }
    if (done == total) break;
    for (int b = 0; b < nblk; ++b) {
      // release waves whose live lanes all wait at a wave op
      for (int w = 0; w < nw; ++w) {
        int lo = b * n + w * 64, hi = std::min(b * n + n, lo + 64), waiting = 0, live = 0;
        for (int i = lo; i < hi; ++i) {
          if (fibers[i].st != DONE) ++live;
          if (fibers[i].st == WAIT_WAVE) ++waiting;
        }
        if (live && waiting == live) {
          waves[(size_t)b * nw + w].gen++;
          for (int i = lo; i < hi; ++i) if (fibers[i].st == WAIT_WAVE) fibers[i].st = RUN;
          progressed = true;
        }
      }
      // release the block barrier when every live thread of the block waits on it
      int live = 0, atbar = 0;
      for (int i = b * n; i < b * n + n; ++i) {
        if (fibers[i].st != DONE) ++live;
        if (fibers[i].st == WAIT_BLOCK) ++atbar;
      }
      if (live && atbar == live) {
        for (int i = b * n; i < b * n + n; ++i) if (fibers[i].st == WAIT_BLOCK) fibers[i].st = RUN;
        progressed = true;
      }
    }
    if (!progressed) {
      fprintf(stderr, "hipemu: deadlock (divergent barrier / partial-wave shuffle) in block (%u,%u,%u)\n", g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
      abort();
    }
  }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  g_gridDim = grid;
  g_blockDim = block;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) run_blocks(block, {uint3_emu{x, y, z}}, shmem, body);
}

void spin_yield() { yield(RUN); }
static int g_tpb() { return g_threads_per_block > 0 ? g_threads_per_block : 1; }
static int wave_of_cur() {
  const int n = g_tpb(), nw = (n + 63) / 64;
  return (cur / n) * nw + ((cur % n) >> 6);
}
void* dyn_shared() { return g_dyn_pool[cur / g_tpb()].data(); }

void enqueue(hipStream_t st, std::function<void()> fn) {
  if (st && st->capturing) st->g->nodes.push_back(std::move(fn));
  else fn();
}

}  // namespace hipemu

// ---------------------------------------------------------------------------------- runtime API
namespace {
std::map<void*, std::pair<void*, size_t>> g_allocs;  // user ptr -> (mapping base, mapping size)
hipemu_stream g_null_stream;
}

hipError_t hipMalloc(void** p, size_t n) {
  const size_t page = 4096;
  size_t body = (n + 15) / 16 * 16;
  size_t pages = (body + page - 1) / page * page;
  char* base = (char*)mmap(nullptr, pages + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (base == MAP_FAILED) return hipErrorOutOfMemory;
  mprotect(base + pages, page, PROT_NONE);  // reads/writes past the end fault immediately
  char* user = base + pages - body;
  memset(base, 0xFF, pages);  // poison: uninitialised floats read as NaN
  g_allocs[user] = {base, pages + page};
  *p = user;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  auto it = g_allocs.find(p);
  if (it == g_allocs.end()) return hipErrorInvalidValue;
  munmap(it->second.first, it->second.second);
  g_allocs.erase(it);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  hipemu::enqueue(st, [=]() { memcpy(d, s, n); });
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  hipemu::enqueue(st, [=]() { memset(d, v, n); });
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hipemu_stream(); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipemu_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
// marker export: "device" memory of this build is host memory (native.NativeLib.host_emulated)
extern "C" int hipemu_host_memory(void) { return 1; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "hipemu (host fibers)");
  strcpy(p->gcnArchName, "hipemu");
  p->multiProcessorCount = 4;
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
  e->g = (st && st->capturing) ? st->g : nullptr;
  if (st && st->capturing) return hipSuccess;
  e->t = now_ms();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  // cross-stream capture: waiting on an event recorded in a capturing stream joins that capture
  if (s && e && e->g && !s->capturing) { s->capturing = true; s->g = e->g; e->g->joined.push_back(s); }
  return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
  if (!s) return hipErrorInvalidValue;
  s->capturing = true;
  s->g = new hipemu_graph();
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
  for (auto* j : s->g->joined) { j->capturing = false; j->g = nullptr; }
  s->g->joined.clear();
  s->capturing = false; *g = s->g; s->g = nullptr; return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { *e = new hipemu_graph(*g); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { for (auto& f : e->nodes) f(); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
