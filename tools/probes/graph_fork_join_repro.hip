// Stand-alone reproduction attempt of the hipStreamEndCapture crash seen when the training step is captured WITH the 3x3x3 branch stream
// (dose_prediction_amd/config.py set_capture_side_streams).  No torch: plain HIP runtime, one trivial kernel.
//
// The step's stream graph during a capture on stream A (what the package and the autograd engine do):
//   * branch stream B: forked from A (event on A, B waits), a few kernels on B beside kernels on A, joined (event on B, A waits) -- once per
//     multi-scale block in the forward pass and again (by the autograd engine) in the backward pass: ~30 fork / join pairs per step;
//   * weight-gradient stream W: forked from WHICHEVER stream runs the convolution's backward node -- A or B -- with a fresh event per fork
//     (~80 per step), never joined to its origin but only to A, ONCE, at the end of the backward pass ("cross join": W's work forked from B
//     is joined into A directly, B itself is joined into A separately);
//   * transformer stream V: one long fork from A, joined near the end.
// usage: graph_fork_join_repro <pattern> <pairs> <mode> [reuse_events]
//   pattern 0: A <-> B fork / join pairs only
//   pattern 1: + W forked from A each pair, joined to A once at the end
//   pattern 2: + W forked from B (inside the pair), joined to A once at the end           (the shipped configuration's shape)
//   pattern 3: pattern 2, but W is ALSO joined to B before B joins A (every fork joins its origin)
//   pattern 4: pattern 2 with the join of B into A BEFORE W's work is enqueued on W from B's event (B's join does not cover W)
//   mode 0 global, 1 thread-local, 2 relaxed;  reuse_events 1: one event object per edge kind re-recorded every pair (default: fresh events)
//   pattern 98: an explicit edge list (argv[5], e.g. "01 12 02 21"); pattern 99: a random event graph over four streams (argv[4] = seed).
// RESULT (round 6): patterns 0-9 all capture fine; **pattern 98 with "01 12 02 21" dies in hipStreamEndCapture** -- unbounded recursion of
// hip::Stream::EndCapture() over a cycle in the runtime's parallel-capture-stream lists (docs/experiments/r06_graph_capture_branch_stream.md).
// Prints one line: "<pattern> <pairs> <mode> <reuse>: nodes=<n> OK" or the failing call; a crash shows as the shell's exit status.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 2; } } while (0)

__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }

int main(int argc, char** argv) {
  const int pattern = argc > 1 ? atoi(argv[1]) : 2, pairs = argc > 2 ? atoi(argv[2]) : 30, mode = argc > 3 ? atoi(argv[3]) : 0;
  const int reuse = argc > 4 ? atoi(argv[4]) : 0;
  hipStream_t A, B, W;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&W, hipStreamNonBlocking));
  float* buf; const int n = 1 << 16;
  CK(hipMalloc(&buf, 4 * n * sizeof(float))); CK(hipMemset(buf, 0, 4 * n * sizeof(float)));
  std::vector<hipEvent_t> evs;
  hipEvent_t shared[4];
  for (auto& e : shared) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto ev = [&](int kind) -> hipEvent_t {
    if (reuse) return shared[kind];
    hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) exit(3);
    evs.push_back(e); return e;
  };
  auto launch = [&](auto s_, int slot) { hipStream_t s = (hipStream_t)s_; hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, s, buf + slot * n, n); };
  const hipStreamCaptureMode cm = mode == 0 ? hipStreamCaptureModeGlobal : mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
  CK(hipStreamBeginCapture(A, cm));
  bool w_used = false;
  for (int i = 0; i < (pattern <= 4 ? pairs : 0); i++) {
    hipEvent_t f = ev(0);
    CK(hipEventRecord(f, A)); CK(hipStreamWaitEvent(B, f, 0));                 // fork A -> B
    launch(B, 1);
    if (pattern == 1) { hipEvent_t g = ev(1); CK(hipEventRecord(g, A)); CK(hipStreamWaitEvent(W, g, 0)); launch(W, 2); w_used = true; }
    if (pattern == 2 || pattern == 3) { hipEvent_t g = ev(1); CK(hipEventRecord(g, B)); CK(hipStreamWaitEvent(W, g, 0)); launch(W, 2); w_used = true; }
    hipEvent_t g4 = nullptr;
    if (pattern == 4) { g4 = ev(1); CK(hipEventRecord(g4, B)); }
    launch(A, 0);
    launch(B, 1);
    if (pattern == 3) { hipEvent_t h = ev(2); CK(hipEventRecord(h, W)); CK(hipStreamWaitEvent(B, h, 0)); }
    hipEvent_t j = ev(3);
    CK(hipEventRecord(j, B)); CK(hipStreamWaitEvent(A, j, 0));                 // join B -> A
    if (pattern == 4) { CK(hipStreamWaitEvent(W, g4, 0)); launch(W, 2); w_used = true; }
    launch(A, 0);
  }
  if (w_used) { hipEvent_t j = ev(2); CK(hipEventRecord(j, W)); CK(hipStreamWaitEvent(A, j, 0)); }      // the one join of W, into A
  // patterns 5-9 (round 6, after tools/probes/graph_capture_bisect.py: the crash needs the backward pass AND the transformer stream V AND the branch
  // stream B): two streams forked from A with CROSS edges between them, each joined into A on its own; `pairs` cross edges
  if (pattern >= 5 && pattern <= 9) {
    hipStream_t V = W;
    hipEvent_t f = ev(0); CK(hipEventRecord(f, A)); CK(hipStreamWaitEvent(V, f, 0));                     // fork A -> V (long branch)
    launch(V, 2);
    hipEvent_t f2 = ev(0); CK(hipEventRecord(f2, A)); CK(hipStreamWaitEvent(B, f2, 0));                  // fork A -> B
    launch(B, 1); launch(A, 0);
    for (int i = 0; i < pairs; i++) {
      if (pattern == 5 || pattern == 6) { hipEvent_t e = ev(1); CK(hipEventRecord(e, V)); CK(hipStreamWaitEvent(B, e, 0)); launch(B, 1); launch(V, 2); }   // V -> B
      if (pattern == 6 || pattern == 7) { hipEvent_t e = ev(2); CK(hipEventRecord(e, B)); CK(hipStreamWaitEvent(V, e, 0)); launch(V, 2); launch(B, 1); }   // B -> V
      if (pattern == 8) {      // B joins A and is forked again every round while V keeps waiting on B's events (the engine's per-node syncs)
        hipEvent_t e = ev(2); CK(hipEventRecord(e, B)); CK(hipStreamWaitEvent(V, e, 0)); launch(V, 2);
        hipEvent_t j = ev(3); CK(hipEventRecord(j, B)); CK(hipStreamWaitEvent(A, j, 0)); launch(A, 0);
        hipEvent_t g = ev(0); CK(hipEventRecord(g, A)); CK(hipStreamWaitEvent(B, g, 0)); launch(B, 1);
      }
      if (pattern == 9) {      // V waits on an event of B that was recorded BEFORE B's join into A, AFTER that join
        hipEvent_t e = ev(2); CK(hipEventRecord(e, B));
        hipEvent_t j = ev(3); CK(hipEventRecord(j, B)); CK(hipStreamWaitEvent(A, j, 0)); launch(A, 0);
        CK(hipStreamWaitEvent(V, e, 0)); launch(V, 2);
        hipEvent_t g = ev(0); CK(hipEventRecord(g, A)); CK(hipStreamWaitEvent(B, g, 0)); launch(B, 1);
      }
    }
    hipEvent_t jb = ev(3); CK(hipEventRecord(jb, B)); CK(hipStreamWaitEvent(A, jb, 0));
    hipEvent_t jv = ev(3); CK(hipEventRecord(jv, V)); CK(hipStreamWaitEvent(A, jv, 0));
    launch(A, 0);
  }
  // pattern 99 (round 6, after rocgdb showed the crash to be UNBOUNDED RECURSION inside hip::Stream::EndCapture -- a cycle in the runtime's
  // "parallel capture streams" lists): a random event graph over four streams; `pairs` = number of record/wait edges, argv[4] (reuse) = seed.
  // Prints the edge list first, so that a crashing seed documents itself.
  // pattern 97 = pattern 99 with EVERY stream first joined to the capture by an event of the origin (0>1 0>2 0>3), then random edges: the candidate
  // work-around (no stream ever joins the capture through a non-origin stream's event)
  if (pattern == 99 || pattern == 97) {
    hipStream_t X; CK(hipStreamCreateWithFlags(&X, hipStreamNonBlocking));
    hipStream_t S[4] = {A, B, W, X}; bool cap[4] = {true, false, false, false};
    unsigned rng = 12345u + 7919u * (unsigned)reuse;
    auto rnd = [&](int n) { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 16) % (unsigned)n); };
    printf("edges:");
    if (pattern == 97) for (int k = 1; k < 4; k++) {
      hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      CK(hipEventRecord(e, A)); CK(hipStreamWaitEvent(S[k], e, 0)); cap[k] = true; launch(S[k], k & 3); printf(" 0>%d", k);
    }
    for (int i = 0; i < pairs; i++) {
      int src; do { src = rnd(4); } while (!cap[src]);
      int dst; do { dst = rnd(4); } while (dst == src);
      hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      launch(S[src], src & 3);
      CK(hipEventRecord(e, S[src])); CK(hipStreamWaitEvent(S[dst], e, 0));
      cap[dst] = true;
      launch(S[dst], dst & 3);
      printf(" %d>%d", src, dst);
    }
    for (int k = 3; k >= 1; k--) if (cap[k]) { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventRecord(e, S[k])); CK(hipStreamWaitEvent(A, e, 0)); printf(" %d>0", k); }
    launch(A, 0);
    printf("\n"); fflush(stdout);
  }
  // pattern 98: an explicit edge list in argv[5] ("01 12 21": record on the first stream, wait on the second), then every capturing stream joins 0
  if (pattern == 98 && argc > 5) {
    hipStream_t X; CK(hipStreamCreateWithFlags(&X, hipStreamNonBlocking));
    hipStream_t S[4] = {A, B, W, X}; bool cap[4] = {true, false, false, false};
    for (const char* p = argv[5]; *p; p++) {
      if (*p < '0' || *p > '3' || p[1] < '0' || p[1] > '3') continue;
      const int src = *p - '0', dst = p[1] - '0'; p++;
      if (!cap[src] || src == dst) { printf("invalid edge %d>%d\n", src, dst); return 4; }
      hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      launch(S[src], src & 3);
      CK(hipEventRecord(e, S[src])); CK(hipStreamWaitEvent(S[dst], e, 0));
      cap[dst] = true;
      launch(S[dst], dst & 3);
    }
    const bool nojoin = argc > 6;
    for (int k = 3; k >= 1; k--) if (cap[k] && !nojoin) { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventRecord(e, S[k])); CK(hipStreamWaitEvent(A, e, 0)); }
    launch(A, 0);
  }
  hipGraph_t graph;
  CK(hipStreamEndCapture(A, &graph));
  size_t nodes = 0; CK(hipGraphGetNodes(graph, nullptr, &nodes));
  hipGraphExec_t exec;
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, A)); CK(hipGraphLaunch(exec, A));
  CK(hipStreamSynchronize(A));
  float h[4];
  for (int s = 0; s < 3; s++) CK(hipMemcpy(&h[s], buf + s * n, sizeof(float), hipMemcpyDeviceToHost));
  printf("pattern %d pairs %d mode %d reuse %d: nodes=%zu OK (counts %.0f %.0f %.0f)\n", pattern, pairs, mode, reuse, nodes, h[0], h[1], h[2]);
  return 0;
}
