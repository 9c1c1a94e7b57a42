// tools/emu/emu_runtime.cpp -- runtime of the host-side kernel-logic simulator (TEST ONLY).
// See tools/emu/hip/hip_runtime.h for what this is and what it is not.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>

extern "C" void bcpemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl bcpemu_switch
.type bcpemu_switch,@function
bcpemu_switch:
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
.size bcpemu_switch, .-bcpemu_switch
)");

namespace bcpemu {

thread_local BlockCtx* g_ctx = nullptr;
thread_local uint3_ g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

static const size_t kStack = 96 * 1024;

struct WaveState { unsigned arrived = 0, gen = 0, alive = 0; };
static thread_local std::vector<WaveState>* g_wstate = nullptr;

static inline void yield_to_sched() {
  BlockCtx* c = g_ctx;
  Fiber& f = c->fibers[c->cur];
  bcpemu_switch(&f.sp, c->sched_sp);
}

int lane_id() {
  BlockCtx* c = g_ctx;
  return c->cur & 63;
}
WaveScratch& my_wave() {
  BlockCtx* c = g_ctx;
  return c->waves[c->cur >> 6];
}
unsigned next_phase() {
  BlockCtx* c = g_ctx;
  unsigned p = c->lane_phase[c->cur];
  c->lane_phase[c->cur] = p ^ 1u;
  return p;
}
void* dyn_lds() { return g_ctx->dyn; }

void block_sync() {
  BlockCtx* c = g_ctx;
  unsigned gen = c->bar_gen;
  if (++c->bar_arrived >= c->alive) {
    c->bar_arrived = 0;
    c->bar_gen++;
    return;
  }
  while (c->bar_gen == gen) yield_to_sched();
}

void wave_sync() {
  BlockCtx* c = g_ctx;
  WaveState& w = (*g_wstate)[c->cur >> 6];
  unsigned gen = w.gen;
  if (++w.arrived >= w.alive) {
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == gen) yield_to_sched();
}

f32x4 mfma_16x16x4(float a, float b, f32x4 cacc) {
  // v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15];
  // D: col = l&15, row = (l>>4)*4 + r.   D = k-ordered fmaf chain (guide section 3).
  WaveScratch& w = my_wave();
  unsigned p = next_phase();
  int l = lane_id();
  w.a[p][l] = a;
  w.b[p][l] = b;
  wave_sync();
  int col = l & 15, rg = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = rg * 4 + r;
    float acc = cacc[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w.a[p][k * 16 + row], w.b[p][k * 16 + col], acc);
    cacc[r] = acc;
  }
  return cacc;
}

f32x16 mfma_32x32x2(float a, float b, f32x16 cacc) {
  // v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
  // D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16).
  WaveScratch& w = my_wave();
  unsigned p = next_phase();
  int l = lane_id();
  w.a[p][l] = a;
  w.b[p][l] = b;
  wave_sync();
  int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = cacc[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(w.a[p][k * 32 + row], w.b[p][k * 32 + col], acc);
    cacc[r] = acc;
  }
  return cacc;
}

f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 cacc) {
  // v_mfma_f32_16x16x32_bf16: lane l supplies A[i=l&15][k=(l>>4)*8+j] and B[k=(l>>4)*8+j][col=l&15], j = 0..7;
  // D: col = l&15, row = (l>>4)*4 + r.  bf16 x bf16 products are exact in fp32; the 32-term sum + C is modelled in
  // fp64 and rounded once (the matrix core keeps more than fp32 inside the dot product; tests compare with tolerances).
  WaveScratch& w = my_wave();
  unsigned p = next_phase();
  int l = lane_id();
  memcpy(w.a16[p][l], &a, 16);
  memcpy(w.b16[p][l], &b, 16);
  wave_sync();
  auto bf = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
  int col = l & 15, rg = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = rg * 4 + r;
    double acc = (double)cacc[r];
    for (int kg = 0; kg < 4; ++kg)
      for (int j = 0; j < 8; ++j) acc += (double)bf(w.a16[p][kg * 16 + row][j]) * (double)bf(w.b16[p][kg * 16 + col][j]);
    cacc[r] = (float)acc;
  }
  return cacc;
}

f32x4 mfma_16x16x32_f16(f16x8_emu a, f16x8_emu b, f32x4 cacc) {
  // v_mfma_f32_16x16x32_f16: the bf16 instruction's lane layout with IEEE half operands (products exact in fp32; dot product + C in
  // fp64, rounded once -- see mfma_16x16x32_bf16)
  WaveScratch& w = my_wave();
  unsigned p = next_phase();
  int l = lane_id();
  memcpy(w.a16[p][l], &a, 16);
  memcpy(w.b16[p][l], &b, 16);
  wave_sync();
  auto hf = [](uint16_t h) { _Float16 f; memcpy(&f, &h, 2); return (double)(float)f; };
  int col = l & 15, rg = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = rg * 4 + r;
    double acc = (double)cacc[r];
    for (int kg = 0; kg < 4; ++kg)
      for (int j = 0; j < 8; ++j) acc += hf(w.a16[p][kg * 16 + row][j]) * hf(w.b16[p][kg * 16 + col][j]);
    cacc[r] = (float)acc;
  }
  return cacc;
}

uint64_t ds_read_tr16_b64(const void* ptr) {
  WaveScratch& w = my_wave();
  unsigned p = next_phase();
  int l = lane_id();
  uint64_t mine;
  memcpy(&mine, ptr, 8);
  w.u[p][l] = mine;
  wave_sync();
  uint64_t out = 0;
  const int g0 = l & ~15, i = l & 15;
  for (int j = 0; j < 4; ++j) {
    const uint64_t src = w.u[p][g0 + 4 * j + (i >> 2)];
    out |= ((src >> (16 * (i & 3))) & 0xFFFFull) << (16 * j);
  }
  return out;
}

static void fiber_entry() {
  BlockCtx* c = g_ctx;
  (*c->body)();
  // fibre finished
  c = g_ctx;
  Fiber& f = c->fibers[c->cur];
  f.done = true;
  c->alive--;
  WaveState& w = (*g_wstate)[c->cur >> 6];
  w.alive--;
  if (w.alive > 0 && w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
  if (c->alive > 0 && c->bar_arrived >= c->alive) { c->bar_arrived = 0; c->bar_gen++; }
  bcpemu_switch(&f.sp, c->sched_sp);
  abort();
}

static void run_block(BlockCtx& c, dim3 block, size_t shmem, const std::function<void()>& body) {
  unsigned n = block.x * block.y * block.z;
  c.nthreads = n;
  c.alive = n;
  c.bar_arrived = 0;
  c.bar_gen = 0;
  c.body = &body;
  if (c.stacks_cap < (size_t)n * kStack) {
    if (c.stacks) munmap(c.stacks, c.stacks_cap);
    c.stacks_cap = (size_t)n * kStack;
    c.stacks = (char*)mmap(nullptr, c.stacks_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (c.stacks == MAP_FAILED) { perror("mmap"); abort(); }
  }
  if (c.dyn_cap < shmem + 64) {
    free(c.dyn);
    c.dyn_cap = shmem + 64;
    c.dyn = (char*)aligned_alloc(64, (c.dyn_cap + 63) / 64 * 64);
  }
  c.fibers.resize(n);
  c.lane_phase.assign(n, 0);
  unsigned nw = (n + 63) / 64;
  c.waves.resize(nw);
  static thread_local std::vector<WaveState> wst;
  wst.assign(nw, WaveState());
  g_wstate = &wst;
  for (unsigned i = 0; i < n; ++i) {
    Fiber& f = c.fibers[i];
    f.done = false;
    f.stack = c.stacks + (size_t)i * kStack;
    f.tid.x = i % block.x;
    f.tid.y = (i / block.x) % block.y;
    f.tid.z = i / (block.x * block.y);
    uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
    uint64_t* s = (uint64_t*)top;
    s[-1] = 0;                          // fake return address of fiber_entry
    s[-2] = (uint64_t)&fiber_entry;     // 'ret' target of the first switch
    for (int k = 3; k <= 8; ++k) s[-k] = 0;
    f.sp = (void*)(s - 8);
    wst[i >> 6].alive++;
  }
  g_ctx = &c;
  while (c.alive > 0) {
    for (unsigned i = 0; i < n; ++i) {
      Fiber& f = c.fibers[i];
      if (f.done) continue;
      c.cur = (int)i;
      g_threadIdx = f.tid;
      bcpemu_switch(&c.sched_sp, f.sp);
    }
  }
  g_ctx = nullptr;
}

// ---------------- persistent worker pool
struct Pool {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> threads;
  uint64_t job_id = 0;
  // job description
  dim3 grid, block;
  size_t shmem = 0;
  const std::function<void()>* body = nullptr;
  std::atomic<uint64_t> next{0};
  uint64_t total = 0;
  unsigned busy = 0;
  bool stop = false;

  void work(BlockCtx& ctx) {
    for (;;) {
      uint64_t b = next.fetch_add(1);
      if (b >= total) break;
      g_blockIdx.x = (unsigned)(b % grid.x);
      g_blockIdx.y = (unsigned)((b / grid.x) % grid.y);
      g_blockIdx.z = (unsigned)(b / ((uint64_t)grid.x * grid.y));
      g_blockDim = {block.x, block.y, block.z};
      g_gridDim = {grid.x, grid.y, grid.z};
      run_block(ctx, block, shmem, *body);
    }
  }
  void worker() {
    BlockCtx ctx;
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || job_id != seen; });
        if (stop) return;
        seen = job_id;
      }
      work(ctx);
      {
        std::unique_lock<std::mutex> lk(mu);
        if (--busy == 0) cv_done.notify_all();
      }
    }
  }
  void ensure() {
    if (!threads.empty()) return;
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = getenv("BCP_EMU_THREADS")) n = (unsigned)atoi(e);
    if (n < 1) n = 1;
    for (unsigned i = 0; i + 1 < n; ++i) threads.emplace_back([this] { worker(); });
  }
};
static Pool* g_pool = nullptr;
static std::mutex g_launch_mu;

void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  std::lock_guard<std::mutex> launch_lock(g_launch_mu);
  if (!g_pool) g_pool = new Pool();  // leaked on purpose (threads detached at exit)
  Pool& P = *g_pool;
  P.ensure();
  static thread_local BlockCtx main_ctx;
  {
    std::unique_lock<std::mutex> lk(P.mu);
    P.grid = grid;
    P.block = block;
    P.shmem = shmem;
    P.body = &body;
    P.total = (uint64_t)grid.x * grid.y * grid.z;
    P.next = 0;
    P.busy = (unsigned)P.threads.size();
    P.job_id++;
  }
  P.cv_job.notify_all();
  P.work(main_ctx);
  {
    std::unique_lock<std::mutex> lk(P.mu);
    P.cv_done.wait(lk, [&] { return P.busy == 0; });
  }
}

}  // namespace bcpemu
