// fh_share.hip.hpp — device-scope work sharing between the persistent workgroups of the solve kernels (gfx950).
//
// Why: one genNewTraj() is a sequence of branch-and-bound trees (one per factor trial).  Almost all of them have a handful of
// nodes, a few have 10^2..10^4, and a tree explored by ONE wavefront pins that wavefront for milliseconds while the other
// ~2000 resident wavefronts have run out of problems (round-1 profile: 7.7 / 16.2 ms per launch for ~3 ms of bulk work).
// Gurobi explores one tree with all its threads behind m.optimize() (/root/reference/faster/src/solverGurobi.cpp:566,
// Threads = 0, faster/param/faster.yaml:41); this file is the counterpart: a wavefront that has run out of fresh problems
// takes over untried sibling subtrees of a tree that is still being explored elsewhere.
//
// Protocol (/opt/skills/guides/cdna_hip_programming.md G16, form R1): EVERY word another workgroup may read is written with an
// 8-byte agent-scope store (write-through, `sc1`) and read with an agent-scope load (served by L2, never by a stale L1 line);
// a producer drains its stores (`s_waitcnt vmcnt(0)`) before it raises the flag / sequence number / counter that publishes
// them.  No release fence is used anywhere: `buffer_wbl2` would write back the whole XCD's dirty L2 — megabytes of other
// workgroups' node snapshots — on every hand-off (measured: a 32768-problem launch went from 8 ms to 170 ms).
//   * ShareCtl        one per context: ticket counter, hand-off counters, units done, error word;
//   * hand-off ring   a workgroup that runs out of problems draws a WAIT TICKET t (one atomic add) and from then on polls only
//                     its own word seqs[t % FH_QCAP] — no shared hot word, no compare-and-swap storm when a frame appears
//                     (the first version let every poller race for a queue head: a 32768-problem launch became 2-3x slower).
//                     A donor publishes frame number p = q_tail only while p < wait_ticket, so every published frame has a
//                     committed taker, and a taker whose frame never comes leaves when all units are done.  A frame is the
//                     parent node's dual active-set state (the same snapshot block the owner keeps per tree level), the
//                     ordered list of untried children, the partial assignment and the DFS key of the parent;
//   * ShareRec        one per problem that has given work away: incumbent (trial, cost, DFS key, jerks, assignment) under a
//                     spin lock, the number of outstanding parts of the problem's tree, accumulated statistics.
// Nobody ever waits for anybody: the part that brings `pending` to zero writes the problem's result, whichever workgroup that
// is.  Results do not depend on who explored what: the answer is the lexicographic minimum of (trial, cost, DFS key) over all
// leaves — the first factor with a feasible trajectory, its cheapest leaf, and among equally cheap leaves the first in
// depth-first order, which is what the sequential search returns — and the jerks of a leaf depend only on the path from the
// root of its trial (every node continues from a bit-exact copy of its parent's factorisation).
//
// Every spin is bounded (FH_SPIN_LIMIT / a wall-clock watchdog): a protocol failure raises ShareCtl::error, all workers
// leave, and the host reports FH_ERR_DEVICE instead of hanging the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fh {

#define FH_QCAP 1024          // task slots in the ring (power of two)
#define FH_NRECS 4096         // share records per launch (problems that gave work away)
#define FH_MAX_GRID 4096      // workgroups of a launch whose first tickets can be dealt (claims[])
#define FH_SPIN_LIMIT (1u << 22)
#define FH_WATCHDOG_TICKS (20ull * 100000000ull)  // 20 s of the 100 MHz s_memrealtime clock: a hungry worker gives up

struct ShareCtl {
  // ---- line 0: polled by hungry workers; zeroed before every launch ----
  unsigned int done;         // units (problems, or pairs) whose final result has been written
  unsigned int done_iters;   // active-set iterations of those units (the running mean decides who may publish ahead: giant_factor)
  unsigned int error;        // != 0: protocol failure / watchdog, everybody leaves        } one aligned 8-byte pair: the busy
  unsigned int interrupted;  // a worker has seen the host's abort word / the deadline    } workers read both with one load
  unsigned int pad0[12];
  // ---- line 1: zeroed before every launch ----
  unsigned long long ticket;  // next fresh unit
  unsigned int rec_next;      // share records handed out
  unsigned int donated, stolen, q_full, rec_full, lock_spins, max_fill;  // statistics
  unsigned int exited;        // workgroups that have left the kernel: the last one re-initialises this block for the next launch
  unsigned int started;       // workgroups that have begun: publishing ahead is for a launch that has the device to itself (all begun)
  unsigned int steal_tries;   // workgroups that ran out of tickets and looked for an unclaimed dealt chunk (ShareArgs::claims, steal_chunk)
  unsigned int pad1[4];
  // ---- line 2: the hand-off counters (zeroed before every launch; written only when a worker runs out of problems or a frame
  // is published, read by the busy workers every few nodes) ----
  unsigned int wait_ticket;  // wait tickets drawn: takers committed to frame numbers 0 .. wait_ticket-1   } one aligned 8-byte
  unsigned int q_tail;       // frames published (or being published): always <= wait_ticket            } pair, read with one load
  unsigned long long pad2[7];
  // ---- line 3: -DFH_SHARE_PROFILE builds only: 100 MHz ticks spent in / number of [0,1] look_around, [2,3] donate,
  // [4,5] waiting for a frame (successful waits), [6,7] copying a frame out of its slot ----
  unsigned long long prof[8];
  // ---- line 4: [0,1] staging + trial set-up of a taken frame until its first node, [2,3] searching taken frames (ticks, nodes),
  // [4,5] finish_part (ticks, count), [6] ticks of workers between "tickets exhausted" and leaving, [7] workers that left ----
  unsigned long long prof2[8];
  // ---- line 5: what the host reads after a launch (copied here by the last workgroup before it resets lines 0-2) ----
  unsigned int report[16];  // donated, stolen, q_full, rec_full, records used, error, interrupted, trial frames
};
static_assert(sizeof(ShareCtl) == 384, "six 64-byte lines");

// The incumbent of a shared problem is ordered by (factor trial, cost, DFS key): genNewTraj() returns the FIRST factor of the
// window with a feasible trajectory (solverGurobi.cpp:445-446), so a leaf of an earlier trial beats every leaf of a later one, and
// the trials of a problem that has proved hard may run at the same time (the trial loop is the outermost level of the tree).
// Lock-free readers prune with one monotonically decreasing word, the RANK = trial << 51 | (cost bits rounded UP) >> 13: a
// rounded-up cost can only prune less; exact (cost, key) comparisons are made under the lock.
#define FH_RANK_SHIFT 51
#define FH_RANK_NONE (~0ull)
__device__ __forceinline__ unsigned long long rank_pack(int trial, double cost) {
  const unsigned long long b = ((unsigned long long)__double_as_longlong(cost) + 0x1fffull) >> 13;
  return ((unsigned long long)(unsigned)trial << FH_RANK_SHIFT) | b;
}
__device__ __forceinline__ int rank_trial(unsigned long long r) { return (int)(r >> FH_RANK_SHIFT); }
__device__ __forceinline__ double rank_cost_up(unsigned long long r) {
  return __longlong_as_double((long long)((r & ((1ull << FH_RANK_SHIFT) - 1ull)) << 13));
}

struct ShareRec {  // 512 B
  unsigned int lock;
  int pending;                    // outstanding parts of the problem's tree (the owner's part + frames given away, of every trial)
  unsigned long long inc_rank;    // FH_RANK_NONE: no feasible leaf yet
  unsigned long long inc_cost;    // exact cost of the incumbent (double bits), its DFS key, its factor and step: under the lock
  unsigned long long inc_key;
  double inc_f, inc_h;
  int nodes, iters;               // accumulated by finished parts
  unsigned int last_status;       // status to report if no factor is feasible: FH_ST_INTERRUPTED, or a limit hit in the LAST trial
  unsigned int limit_kind;        // the limit (FH_ST_NODE_LIMIT / FH_ST_ITER_LIMIT) some wavefront ran into in a trial of this problem
  unsigned long long flops;       // accumulated flop estimate
  unsigned long long limited;     // bit min(t, 63): a wavefront hit a node / iteration limit in trial t (its leaves cannot win)
  double x[48];
  unsigned long long assign_lo, assign_hi;  // incumbent assignment, one byte per segment
  unsigned int pad[8];
};
static_assert(sizeof(ShareRec) == 512, "share record layout");

// A task = 32 header words + the snapshot block.  Header words (all 8 bytes, written write-through by one lane):
enum { TH_REC_B = 0,      // rec | b << 32 (rec = -1: empty frame)
       TH_PHASE_DEPTH,    // phase | depth0 << 32
       TH_KEY,            // DFS key of the parent
       TH_H, TH_F, TH_BASE,   // doubles: step of the trial, its factor, max(dt_initial, 2 DC)
       TH_TRIALS_SEG,     // trials | seg << 32
       TH_CNT_NEXT,       // cnt | next << 32
       TH_Q_QE,           // q_saved | qe << 32
       TH_ORDER, TH_ASSIGN_LO, TH_ASSIGN_HI,
       TH_KIND,           // 0: a stack frame of a branch-and-bound tree (snapshot follows); 1: the remaining factor trials of a problem,
                          //    starting with trial number `trials` - 1 at factor TH_F (no snapshot: a trial starts from its root)
       TH_WORDS = 32 };
struct TaskHdr {
  unsigned long long w[TH_WORDS];
};
static_assert(sizeof(TaskHdr) == 256, "task header layout");

struct ShareArgs {
  ShareCtl* ctl;
  unsigned long long* seqs;   // [FH_QCAP] Vyukov sequence numbers, seqs[i] = i after fh_create
  unsigned char* slots;       // [FH_QCAP] x slot_stride bytes: TaskHdr + snapshot
  unsigned long long slot_stride;
  ShareRec* recs;             // [FH_NRECS]
  const unsigned int* host_abort;  // mapped host word written by StopExecution(); may be null
  unsigned long long deadline_ticks;  // s_memrealtime ticks (100 MHz) after which workers stop; 0 = none
  int enabled;                // 0: no frames are given away (workers leave when the tickets are exhausted)
  int total_units;
  int max_hungry;             // at most this many workgroups poll the queue; the others leave when the tickets are exhausted
  int min_nodes;              // a problem gives work to an idle workgroup only after this many branch-and-bound nodes (all trials so far): small trees
                              // are cheaper to finish than to hand over (a hop costs about as much as 2-3 nodes)
  int backlog;                // frames that may be published AHEAD of the takers: a problem that has proved hard does not have to wait
                              // for the fresh problems to run out before it gets help — every workgroup looks for a pending frame
                              // before it draws its next problem (the hard problems started early otherwise finish last, alone)
  int giant_nodes;            // ... but only a problem that has already needed this many nodes publishes ahead of the takers
  int giant_factor;           // ... or this many times the mean number of active-set iterations of the units finished so far (0: off)
  int child_bound;            // 1: a child whose one-row dual bound at the parent already loses against the incumbent is not visited (fh_sched.child_bound)
  int compact_results;        // 1: only the coefficient rows the kernel is built for are written (fh_sched.compact_results)
  // the fused pair kernel: the safe problem of a pair lives in the LDS of the wavefront that solves it; its record and rows are written to
  // memory at the hand-off (pair_outputs: fh_sched.pair_outputs) or when the problem is first shared with another workgroup (otherwise)
  int pair_outputs;
  int look_mask;              // a tree looks around when (its node count & look_mask) == 0 (fh_sched.look_every - 1; a shared problem uses look_mask >> 1)
  unsigned int* claims;       // [FH_MAX_GRID] claims[w] != 0: the chunk of tickets dealt to workgroup w has been taken — by w itself when it
                              // started, or by a workgroup that ran out of tickets before w had started (null: nothing is dealt)
  const fh_problem* whole;    // [n] the whole problems of the launch
  const fh_face* wfaces;      // their rows
  fh_problem* safe;           // [n] the safe problems: the caller's templates + (if written) x0, n_poly, face_off, face_begin
  fh_face* sfaces;            // their rows, at the whole problem's face_begin
  double shrink, r_margin;
};

#define FH_AGENT __HIP_MEMORY_SCOPE_AGENT
#ifdef FH_SHARE_PROFILE
#define FH_SP_T0() const unsigned long long sp_t0__ = __builtin_amdgcn_s_memrealtime()
#define FH_SP_ADD(arr, slot, extra)                                                                                          \
  do {                                                                                                                      \
    if (threadIdx.x == 0) {                                                                                                 \
      __hip_atomic_fetch_add(&sa.ctl->arr[slot], __builtin_amdgcn_s_memrealtime() - sp_t0__, __ATOMIC_RELAXED, FH_AGENT);   \
      __hip_atomic_fetch_add(&sa.ctl->arr[(slot) + 1], (unsigned long long)(extra), __ATOMIC_RELAXED, FH_AGENT);            \
    }                                                                                                                       \
  } while (0)
#else
#define FH_SP_T0()
#define FH_SP_ADD(arr, slot, extra)
#endif
template <typename T>
__device__ __forceinline__ T ald(T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, FH_AGENT); }
template <typename T>
__device__ __forceinline__ void ast(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, FH_AGENT); }
template <typename T>
__device__ __forceinline__ T aadd(T* p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, FH_AGENT); }

// producer side of a hand-off: the write-through stores of this wavefront issued so far have reached L2 / the fabric
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// consumer side, once per hand-off: drops this CU's L1 lines (plain loads of data another workgroup wrote and released, e.g. the
// safe problem of a pair); does not touch L2
__device__ __forceinline__ void acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// 8-byte write-through store / L1-bypassing load of shared payload
__device__ __forceinline__ void wt_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, FH_AGENT);
}
__device__ __forceinline__ void wt_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, FH_AGENT); }
__device__ __forceinline__ double cc_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, FH_AGENT));
}
__device__ __forceinline__ unsigned long long cc_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, FH_AGENT); }

__device__ __forceinline__ unsigned long long wall_ticks() { return __builtin_amdgcn_s_memrealtime(); }

// ---- the hand-off ring.  All functions are called by ONE lane. ----
// seqs[i] == p      : slot i is free for frame number p (p % FH_QCAP == i)
// seqs[i] == p + 1  : frame p is in the slot, payload complete
// seqs[i] == p + FH_QCAP : the taker of frame p has copied it out; the slot is free for frame p + FH_QCAP
// Reserve the next frame number for writing — ONE attempt: ~0ull if no uncommitted taker is waiting, the ring is full, or
// another donor took the number (when the fresh problems run out, every busy workgroup sees the new takers at its next
// look-around: a retry loop here made ~2000 donors hammer q_tail for 512 numbers, 600 us per donation).
__device__ inline unsigned long long q_reserve(const ShareArgs& sa, int backlog = -1) {  // backlog < 0: the launch's (sa.backlog)
  const unsigned long long both = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->wait_ticket));
  const unsigned int waiters = (unsigned int)both, pos = (unsigned int)(both >> 32);
  if (pos >= waiters + (unsigned int)(backlog < 0 ? sa.backlog : backlog)) return ~0ull;  // enough frames are pending already
  if (ald(&sa.seqs[pos & (FH_QCAP - 1)]) != (unsigned long long)pos) return ~0ull;    // slot not released yet (ring full) or the number is gone
  unsigned int expect = pos;
  if (__hip_atomic_compare_exchange_strong(&sa.ctl->q_tail, &expect, pos + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, FH_AGENT)) return pos;
  return ~0ull;
}
// after the payload has been written and drained
__device__ __forceinline__ void q_publish(const ShareArgs& sa, unsigned long long pos) { ast(&sa.seqs[pos & (FH_QCAP - 1)], pos + 1); }
// has frame number `pos` (this taker's wait ticket) arrived?
__device__ __forceinline__ bool q_arrived(const ShareArgs& sa, unsigned long long pos) { return ald(&sa.seqs[pos & (FH_QCAP - 1)]) == pos + 1; }
// after the payload has been read completely
__device__ __forceinline__ void q_release(const ShareArgs& sa, unsigned long long pos) { ast(&sa.seqs[pos & (FH_QCAP - 1)], pos + FH_QCAP); }

__device__ __forceinline__ TaskHdr* slot_hdr(const ShareArgs& sa, unsigned long long pos) {
  return reinterpret_cast<TaskHdr*>(sa.slots + (pos & (FH_QCAP - 1)) * sa.slot_stride);
}
__device__ __forceinline__ double* slot_snap(const ShareArgs& sa, unsigned long long pos) {
  return reinterpret_cast<double*>(sa.slots + (pos & (FH_QCAP - 1)) * sa.slot_stride + sizeof(TaskHdr));
}

// ---- incumbent lock of a share record (one lane) ----
__device__ inline bool rec_lock(const ShareArgs& sa, ShareRec* r) {
  for (unsigned spin = 0; spin < FH_SPIN_LIMIT; spin++) {
    unsigned int expect = 0u;
    if (__hip_atomic_compare_exchange_strong(&r->lock, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, FH_AGENT)) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  ast(&sa.ctl->error, 3u);
  return false;
}
// (the holder drains its write-through stores first)
__device__ __forceinline__ void rec_unlock(ShareRec* r) { __hip_atomic_store(&r->lock, 0u, __ATOMIC_RELAXED, FH_AGENT); }

}  // namespace fh
