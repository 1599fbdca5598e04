// rt_render_kernel.h — k_render_tiles: the reference's whole per-path loop nest (RenderWorker.hs:53-114,
// Trace.hs:19-105, Light.hs:16-49, BVH.hs:120-146) as ONE persistent wave64 kernel.  Design notes, measurements
// and the history of the numbers quoted here: DESIGN.md sections 5 and 7.
//
// Work items.  A lane is a worker that pulls PATHS: item q <-> (tile of this rank's list, position in the 8x8
// tile, sample).  A wave takes fr.chunk consecutive items at a time from one global atomic counter and hands them to
// its lanes as they fall idle (refill); consecutive items are consecutive samples of one pixel, so a freshly filled
// wave traces 64 nearly identical primary rays.  The RNG is keyed by (pixel, sample); the path's radiance goes to
// samples[q] and k_resolve sums each pixel's samples in sample order (RenderWorker.hs:111-114), so a frame does not
// depend on which lane, wave, launch or GPU traced a path.
//
// Vote-scheduled state machine.  Every lane carries its traversal state across trips of the main loop; each trip
// executes exactly ONE of three bodies, picked by wave-uniform ballots, so a finished lane never waits for the
// slowest ray of its wave:
//     NODE  : lanes at an internal node fetch it (LDS copy of the top levels, else global), test both child boxes,
//             descend into the nearer child, push the other one, or pop           (RT_NODE_UNROLL visits per trip)
//     PRIM  : lanes in a leaf test one primitive (FP64, the reference's operations in its order) and advance / pop
//     SHADE : lanes whose query finished run the path transition (hit -> emission, shadow rays light by light,
//             material bounce; miss / depth out -> background, store the sample) and start their next query
// NODE and PRIM are straight-line predicated code: selects instead of nested ifs, no early exits in the triangle
// test, lanes in another state read element 0 and discard the result instead of being masked off, one inline pop
// per step (a lane whose popped entry was culled continues as CUR_POP), spheres / planes behind a wave-uniform branch.
//
// Registers.  Only the traversal state (ray, box-space ray, closest hit, stack pointer: ~40 VGPRs) is live in
// NODE / PRIM.  What only SHADE needs - incoming direction, normal, attenuation, the pending clamp function
// F(x) = clamp(x + a, L, H), sample / bounce counters - is a per-lane RECORD in global memory ([wave][row][lane],
// coalesced 512-B rows), re-read by SHADE; libm bodies (pow, atan2/asin) are kept out of line; SHADE and refill re-read
// their kernel arguments from the kernarg segment (RT_COLD_ARGS).  128 VGPRs: 16 waves per CU in one workgroup.
// The per-lane traversal stack (ref, entry distance) is in LDS, [wave][entry][lane].
//
// State of a lane, in ONE register: cur >= 0 internal node index, CUR_IDLE no item, CUR_DONE query finished (waits
// for SHADE), CUR_POP owes a pop, CUR_FIN / CUR_FIN0 path ended (waits for the refill step, which writes its sample and
// hands the lane a new item), anything else negative = leaf reference ~first_primitive.
// Numerics: emission and each light's term are added to `a` as they arrive, i.e. the reference's
// `e + surf*(l1 + (l2 + ...)) + clamp(..)` is summed in a different order (<= 1 ulp, only with several lights).
#pragma once
#define CUR_IDLE ((int)0x80000000)
#define CUR_DONE ((int)0x80000001)
#define CUR_POP  ((int)0x80000002)   // keeps popping at the next NODE trip
#define CUR_FIN  ((int)0x80000003)   // path ended on a miss: owes its final sample = F(att * background(rd)); written in the next refill step
#define CUR_FIN0 ((int)0x80000004)   // path ran out of depth: the same along the PRIMARY ray (Trace.hs:22), regenerated there
#define CUR_SPECIAL_MAX CUR_FIN0     // a cur above this value (as unsigned) is a leaf reference
#ifndef RT_SPECULATE
#define RT_SPECULATE 1      // 1: a lane that reaches a leaf postpones it and keeps traversing until it holds a second one
#endif
#ifndef RT_NODE_UNROLL
#define RT_NODE_UNROLL 5    // node visits per NODE trip (amortises the vote; with one inline pop: 3: 147.4 ms, 4: 144.3, 5: 142.9, 6: 146.4, 8: 143.9 on C4)
#endif
#ifdef RT_ASM_MARKERS
#define RT_MARK(name) asm volatile("; RTMARK " name ::: "memory")
#else
#define RT_MARK(name) do { } while (0)
#endif
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 6      // traversal-stack entries kept in LDS; deeper entries (4.6 % of the pushes on C4) go to global memory
#endif
#ifndef RT_POPS
#define RT_POPS 1           // inline predicated pops per step; a lane whose popped entries were all culled continues as CUR_POP
#endif
#ifndef RT_PRIM_UNROLL
#define RT_PRIM_UNROLL 1    // primitive tests per PRIM trip
#endif
// FAST mode (the core's own tree: 5.2 node visits and 1.2 primitive tests per ray, no speculation), C4 kernel ms for
// node visits x primitive tests per trip: 5x1 43.5, 4x1 42.9, 5x2 42.8, 4x2 42.2, 3x2 43.5, 4x3 44.2, 6x1 44.1
#ifndef RT_NODE_UNROLL_FAST
#define RT_NODE_UNROLL_FAST 4
#endif
#ifndef RT_PRIM_UNROLL_FAST
#define RT_PRIM_UNROLL_FAST 2
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 24    // idle lanes are given new items once this many wait (or none is live): the refill costs the whole wave
#endif
#ifndef RT_BLOCKS_PER_CU
#define RT_BLOCKS_PER_CU 1    // workgroups resident per CU (each gets 160 KB / RT_BLOCKS_PER_CU of LDS)
#endif
#ifndef RT_BLOCK_THREADS
#define RT_BLOCK_THREADS (256 * RT_MIN_WAVES)   // one workgroup per CU
#endif

#ifndef RT_NODE_WEIGHT
#define RT_NODE_WEIGHT 1    // NODE step runs while n_node * WEIGHT >= n_leaf
#endif
#ifndef RT_LEAF_WEIGHT
#define RT_LEAF_WEIGHT 1
#endif
#ifndef RT_SHADE_NUM
#define RT_SHADE_NUM 5      // run the SHADE body once >= NUM/DEN of the wave's live lanes wait for it (measured 1/2: 159 ms, 5/8: 156.5, 7/8: 167)
#endif
#ifndef RT_SHADE_DEN
#define RT_SHADE_DEN 8
#endif

// rows of the per-lane path record (doubles; 64 lanes per row)
#ifndef RT_REC_ROWS_DEFINED
#define RT_REC_ROWS_DEFINED
// The per-lane path record.  Its hot rows live in LDS next to the traversal stack ([wave][row][lane], doubles);
// the rows only spheres / planes / glass / metal use stay in global memory, followed by the stack's overflow rows.
enum { LR_ATT = 0, LR_FA = 3, LR_FL = 6, LR_FH = 9, LR_SB = 12 /* int2: sample, bounce */, LR_ROWS = 13 };
enum { PR_UDIR = 0, PR_NRM = 3, PR_S = 6 /* Russian roulette (opt-in extension): the weight of everything deeper */, PR_ROWS = 7 };
#define HITP_FRONT ((int)0x80000000)   // lane register hitp: the primitive of the path's current hit | front-face bit

// The cold rows are global memory, and SHADE says so to the compiler: its opaque copy of the row pointer had lost the address
// space, and FLAT loads / stores also count against the LDS counter, so that a wait for the record's LDS rows waited for them too.
typedef __attribute__((address_space(1))) double gdouble;
RT_DEV d3 rec_ld(const gdouble *r, int row) { return mk3(r[row * 64], r[(row + 1) * 64], r[(row + 2) * 64]); }
RT_DEV void rec_st(gdouble *r, int row, d3 v) { r[row * 64] = v.x; r[(row + 1) * 64] = v.y; r[(row + 2) * 64] = v.z; }
RT_DEV d3 rec_ld(const double *r, int row) { return mk3(r[row * 64], r[(row + 1) * 64], r[(row + 2) * 64]); }
RT_DEV void rec_st(double *r, int row, d3 v) { r[row * 64] = v.x; r[(row + 1) * 64] = v.y; r[(row + 2) * 64] = v.z; }
RT_DEV int2 rec_ldi(const double *r, int row) { return *(const int2 *)(r + row * 64); }
RT_DEV void rec_sti(double *r, int row, int a, int b) { *(int2 *)(r + row * 64) = make_int2(a, b); }
#endif


// Instruction-arbitration priority (s_setprio) of a wave while it runs a body.  NODE trips run above the rest: a node visit is a
// chain of dependent LDS / L1 round trips with little arithmetic between them, and a wave that gets its next load out sooner hides
// more of it behind the others' arithmetic (C4: kernel 41.8 -> 41.1 ms at any level above 0; PRIM or SHADE raised instead: no gain or a loss).
#ifndef RT_PRIO_SHADE
#define RT_PRIO_SHADE 0
#endif
#ifndef RT_PRIO_PRIM
#define RT_PRIO_PRIM 0
#endif
#ifndef RT_PRIO_NODE
#define RT_PRIO_NODE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_PRIO(n) do { if (RT_PRIO_SHADE || RT_PRIO_NODE || RT_PRIO_PRIM) __builtin_amdgcn_s_setprio(n); } while (0)
#else
#define RT_PRIO(n) do { } while (0)
#endif

// n / d for n < 2^24 with rcp = 1.0f / d: float estimate, exact after the correction steps
RT_DEV uint32_t fastdiv24(uint32_t n, uint32_t d, float rcp) {
  uint32_t q = (uint32_t)((float)n * rcp);
  int32_t r = (int32_t)(n - q * d);
  while (r < 0) { --q; r += (int32_t)d; }
  while (r >= (int32_t)d) { ++q; r -= (int32_t)d; }
  return q;
}

// gamma correction c ** gamma (Trace.hs:103-105), out of line: the FP64 pow body must not set the kernel's register budget
__device__ RT_NOINLINE double pow_gamma(double c, double g) { return pow(c, g); }

// The two parameter structs are 720 bytes of kernel arguments.  Held in SGPRs for the whole kernel they overflow the
// scalar register file (107 SGPRs were spilled into VGPR lanes and read back with v_readlane inside the NODE and PRIM
// loops).  The cold sections (SHADE, refill) therefore re-read the fields they need from the kernarg segment at the
// point of use - scalar loads that hit the scalar cache - through a pointer the optimiser cannot see through, and the
// values die at the end of the section.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_COLD_ARGS)   // (the host pass only parses the kernel body)
#define RT_KARG_AS __attribute__((address_space(4)))
#define RT_COLD_ARGS()                                                                                    \
  const RT_KARG_AS char *ka_ = (const RT_KARG_AS char *)__builtin_amdgcn_kernarg_segment_ptr();          \
  asm volatile("" : "+s"(ka_));                                                                           \
  const KScene sc = *(const RT_KARG_AS KScene *)ka_;                                                      \
  const KFrame fr = *(const RT_KARG_AS KFrame *)(ka_ + ((sizeof(KScene) + alignof(KFrame) - 1) / alignof(KFrame)) * alignof(KFrame))
#else
#define RT_COLD_ARGS() do { } while (0)
#endif

// An exact tie in FAST mode (this kernel's own tree): would the reference have kept `cand` over `cur`, both hit at the
// same distance?  The reference keeps the one it tests LATER (HitHelpers.hs:14-16: `t <= closest`; BVH.hs:137-141: the
// child visited second receives [min, t1] and wins an equal t).  Same reference leaf: the later element.  Different
// leaves: at their lowest common ancestor the child with the smaller entry distance is visited first, tie => right
// first (BVH.hs:132); a child's clipped interval starts at max(its own slab entry, 0.001) whatever its ancestors
// clipped, because nested boxes give nested slab intervals.  Out of line and rare: duplicated objects (the reference
// lists config triangles twice, SceneParser.hs:67-73) and rays through shared edges or vertices.
__device__ RT_NOINLINE bool tie_candidate_wins(const int32_t *leaf_anc, const int32_t *ref_anc, const int32_t *ref_depth,
                                                             const DNode<double> *ref_nodes, double ox, double oy, double oz,
                                                             double dx, double dy, double dz, int cand, int cur) {
  const int a = leaf_anc[cur], c = leaf_anc[cand];
  if (a == c) return cand > cur;
  int na = a >> 1, nc = c >> 1, side_c = c & 1;
  for (int guard = 0; na != nc && guard < 256; ++guard) {
    if (ref_depth[na] >= ref_depth[nc]) na = ref_anc[na] >> 1;
    else { const int x = ref_anc[nc]; side_c = x & 1; nc = x >> 1; }
  }
  BoxRay<double> br;
  make_boxray(br, mk3(ox, oy, oz), mk3(dx, dy, dz), 0.0f);
  const double *bx = ref_nodes[na].box;
  double eL, eR;
  (void)box_test(br, bx, RT_TMIN, __builtin_huge_val(), eL);
  (void)box_test(br, bx + 6, RT_TMIN, __builtin_huge_val(), eR);
  const bool left_first = eL < eR;
  return side_c == (left_first ? 1 : 0);
}

// FAST: the kernel traverses the render core's own SAH tree (rt_fastbvh.h); primitives carry their index in the
// reference's leaf order, and exact ties are settled as the reference's traversal would (tie_candidate_wins).
// RR: the opt-in Russian roulette extension (rt_params.rr_start_bounce > 0) is a separate instantiation, so that the kernel
// of the reference's estimator carries none of its code (compiled in as a run-time branch it cost 4 % of the frame).
template <typename BT, bool STATS, bool FAST, bool RR = false>
__global__ __launch_bounds__(RT_BLOCK_THREADS, RT_MIN_WAVES) void k_render_tiles(const KScene sc, const KFrame fr) {
  extern __shared__ uint2 lds_all[];                       // [top_nodes] cached nodes (64 B each), then per wave: RT_STACK_LDS stack rows + LR_ROWS record rows
  const int lane = threadIdx.x & 63;
  const float4 *lnodes = (const float4 *)lds_all;          // 4 x float4 per cached FP32 node
  uint2 *stack = lds_all + (size_t)fr.top_nodes * 8 + (size_t)(threadIdx.x >> 6) * (RT_STACK_LDS + LR_ROWS) * 64 + lane;
  double *lrec = (double *)(stack + RT_STACK_LDS * 64);    // this lane's hot record rows (LDS)
  double *rec = fr.fscratch + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * fr.scratch_rows * 64 + lane;
  uint2 *ovf = (uint2 *)(rec + PR_ROWS * 64);              // stack entries RT_STACK_LDS, RT_STACK_LDS + 1, ... of this lane
  // box_precision = 1 (BT = double) is the LITERAL walk: a stacked child is entered iff its exact FP64 entry distance is not beyond
  // the closest hit so far (the reference's [minVal iSecond, t firstHit] is non-empty, BVH.hs:137-138).  The 8-byte stack entry
  // holds that distance rounded down to FP32 (a conservative cull); the exact double goes to a row of its own in global memory.
  double *xst = rec + (size_t)(sizeof(BT) == 8 ? fr.exact_row : 0) * 64;
  const DNode<BT> *nodes = (const DNode<BT> *)sc.nodes;
  const int top_nodes_v = fr.top_nodes;
  const double inf = __builtin_huge_val();
  const BT tminb = tmin_for_boxes<BT>();

  uint32_t st[STAT_COUNT];
#pragma unroll
  for (int i = 0; i < STAT_COUNT; ++i) st[i] = 0;

  if (fr.top_nodes > 0) {                                  // fill the node cache: coalesced 16-B copies, once per workgroup
    const float4 *src = (const float4 *)sc.nodes;
    float4 *dst = (float4 *)lds_all;
    for (int i = threadIdx.x; i < fr.top_nodes * 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  // ---- per-lane work item and traversal state (registers)
  uint32_t pixel = 0;                                      // row*W + col, also the RNG key
  uint32_t item = 0;                                       // index of this path in fr.samples (its sample number is in the record)
  d3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);
  BoxRay<BT> br;
  BT tmaxb = (BT)0;
  double best_t = 0.0;
  int best = -1, cur = CUR_IDLE, sp = 0, li = -1;          // li >= 0: the query is the shadow ray of light li
  // Speculative traversal (FP32 boxes only; the FP64 parity mode visits exactly the reference's nodes): `pend` is a
  // postponed leaf (0 = none).  Leaves are still tested in traversal order (pend before cur), so ties resolve as in
  // the reference; nodes visited before the postponed leaf has narrowed the interval are a superset of the reference's.
  constexpr bool SPEC = (RT_SPECULATE != 0) && sizeof(BT) == 4 && !FAST;   // on the core's own tree rays touch ~1 leaf: postponing it only adds node visits (44.7 -> 43.6 ms without)
  constexpr int NODE_UNROLL = FAST ? RT_NODE_UNROLL_FAST : RT_NODE_UNROLL, PRIM_UNROLL = FAST ? RT_PRIM_UNROLL_FAST : RT_PRIM_UNROLL;
  int pend = 0;
  // the primitive the path last hit (| HITP_FRONT): kept in a register across the shadow queries, so that SHADE can
  // address its normal and material in the FIRST round of loads instead of after reading them from the record
  int hitp = 0;
  // ---- wave-uniform pool of items [pool_next, pool_end)
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;
#ifdef RT_TIMERS
  const unsigned long long dbg_t0 = wall_clock64();
  unsigned long long dbg_texh = 0;
  uint32_t dbg_chunks = 0;
#endif

#define RT_BEGIN_QUERY(q_tmax)                                                     \
  do {                                                                             \
    make_boxray(br, ro, rd, sc.rscene);                                            \
    best_t = (q_tmax); tmaxb = tmax_for_boxes<BT>(best_t);                         \
    best = -1; sp = 0; cur = sc.root_ref; pend = 0;                                \
    if (STATS) st[li >= 0 ? STAT_SHADOW : STAT_CLOSEST]++;                         \
  } while (0)
// Branch-free pop for the lanes with p_ set: RT_POPS entries at most; entries beyond the closest hit are skipped.
#define RT_POP2(p_)                                                                  \
  do {                                                                               \
    _Pragma("unroll") for (int a_ = 0; a_ < RT_POPS; ++a_) {                               \
      const int spn_ = sp - 1;                                                       \
      /* one unsigned min clamps both ends: an empty stack (-1) or an overflow entry reads the last LDS row, ignored below */ \
      uint2 e_ = stack[min((uint32_t)spn_, (uint32_t)(RT_STACK_LDS - 1)) * 64u];       \
      if (__ballot((p_) && spn_ >= RT_STACK_LDS) != 0ull) {          /* rare: the entry is in the overflow rows */ \
        if ((p_) && spn_ >= RT_STACK_LDS) e_ = ovf[(spn_ - RT_STACK_LDS) * 64];      \
      }                                                                              \
      const bool empty_ = sp == 0;                                                   \
      bool culled_ = cull_entry(__uint_as_float(e_.y), tmaxb);                       \
      if constexpr (sizeof(BT) == 8) { if ((p_) && spn_ >= 0) culled_ = xst[spn_ * 64] > best_t; }   \
      cur = (p_) ? (empty_ ? CUR_DONE : (culled_ ? cur : (int)e_.x)) : cur;          \
      sp = ((p_) && !empty_) ? spn_ : sp;                                            \
      p_ = (p_) && !empty_ && culled_;                                               \
    }                                                                                \
    cur = (p_) ? CUR_POP : cur;                                                      \
  } while (0)

  // Path end (Trace.hs:22,25): the sample is F(att * background) with the pending clamp function F of the record.  Lanes do
  // not run this where their path ends (SHADE) but park as CUR_FIN / CUR_FIN0 and run it together at the next refill step:
  // a finished lane would wait there for a new item anyway, and the block - the camera ray regenerated for ray0, the sky
  // lookup - is then executed once for >= RT_REFILL_MIN lanes instead of once per SHADE trip for the ~14 that end in it.
  auto finish_paths = [&]() {
    const bool fin = cur == CUR_FIN || cur == CUR_FIN0;
    if (fin) {
      RT_COLD_ARGS();
      d3 bgdir = rd;
      if (cur == CUR_FIN0) {                             // depth ran out: background along the PRIMARY ray, regenerated
        const int2 sb = rec_ldi(lrec, LR_SB);
        d3 o_;
        camera_ray(fr, (int)(pixel % (uint32_t)fr.W), (int)(pixel / (uint32_t)fr.W), pixel, (uint32_t)sb.x, o_, bgdir);
      }
      const d3 att = rec_ld(lrec, LR_ATT), fa = rec_ld(lrec, LR_FA), fl = rec_ld(lrec, LR_FL), fh = rec_ld(lrec, LR_FH);
      d3 T = mk3(0, 0, 0);
      // (extension) a path the roulette terminated (bounce counter -1) adds NOTHING deeper: no background lookup - 0 x an
      // infinite sky texel would be NaN where the definition (oracle.cpp traceLoop) returns emitted + lit
      const bool dead = RR && rec_ldi(lrec, LR_SB).y < 0;
      if (!dead) {
        T = att * background(sc, bgdir);
        if (RR) T = rec[PR_S * 64] * T;     // (extension) the roulette weight of this depth
      }
      // a path that never hit anything returns its background term UNclamped (Trace.hs:22,25 at the top level): the pending function
      // is the identity then - which clamp(., -inf, +inf) is not for a NaN (see clampd)
      const d3 R = rec_ldi(lrec, LR_SB).y == 0 ? T + fa : clamp3(T + fa, fl, fh);
      double *o_ = fr.samples + 3 * (size_t)item;
      // written once, read once by k_resolve after the launch: non-temporal, so that the stream (5 GB on C4, 51 GB on C5) does not
      // push nodes out of L2 (C4 kernel 41.36 -> 41.12 ms, C5 516.8 -> 512.1; profiles/r03_ab_nt.txt.  The same hint on the sky
      // texel loads costs 1.5 ms on C4: neighbouring samples re-read the same texels, and they should stay cached)
#ifndef RT_SAMPLE_TEMPORAL
      __builtin_nontemporal_store(R.x, o_); __builtin_nontemporal_store(R.y, o_ + 1); __builtin_nontemporal_store(R.z, o_ + 2);
#else
      o_[0] = R.x; o_[1] = R.y; o_[2] = R.z;
#endif
      cur = CUR_IDLE;
    }
  };

  // Gives every idle lane its next item.  All lanes of the wave run this (uniform control flow).
  // A chunk is fr.chunk CONSECUTIVE items (samples of neighbouring pixels keep a wave's rays in the same part of the
  // tree: mixing image regions inside a chunk cost 5 % of throughput).  Its size is set per launch from the items each
  // resident wave will process: the few hundred chunks under the blob (sliver triangles around its pole, five bounces in
  // the gap above the pedestal) cost 9x the median, and with 1024-item chunks whichever waves held them last finished
  // 6.5 ms after the others whatever the launch size - 13.1 instead of 6.4 ms for a rank's 1/8 share of C4.
  auto refill = [&]() {
    RT_COLD_ARGS();
    const uint32_t n_groups = fr.n_items >> 6;             // n_items is a multiple of 64 (64 slots per tile)
    const uint32_t chunk = fr.chunk;
    const float rcp_spp = 1.0f / (float)fr.spp, rcp_tx = 1.0f / (float)fr.tiles_x;
    const bool small_div = fr.spp < (1 << 18) && n_groups < (1u << 24) && (uint32_t)fr.tiles_x * (uint32_t)((fr.H + 7) / 8) < (1u << 24);
    for (;;) {
      const unsigned long long need = __ballot(cur == CUR_IDLE);
      if (need == 0ull) break;
      if (pool_next == pool_end) {                         // pool empty: take the next chunk from the global counter
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(fr.tile_counter, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
#ifdef RT_TIMERS
        if ((unsigned long long)c * chunk >= (unsigned long long)fr.n_items) dbg_texh = wall_clock64(); else ++dbg_chunks;
        if (fr.dbg && lane == 0 && c < 262144u) {            // per chunk: who took it and when
          unsigned long long *e_ = fr.dbg + 4 * (size_t)gridDim.x * (blockDim.x >> 6) + 2 * (size_t)c;
          e_[0] = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); e_[1] = wall_clock64();
        }
#endif
        if ((unsigned long long)c * chunk >= (unsigned long long)fr.n_items) { exhausted = true; break; }
        pool_next = c * chunk;
        pool_end = fr.n_items - pool_next < chunk ? fr.n_items : pool_next + chunk;
      }
      const uint32_t avail = pool_end - pool_next;
      const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));   // my index among the idle lanes
      const uint32_t n_need = (uint32_t)__popcll(need);
      const uint32_t q = pool_next + rank;
      const uint32_t g2 = q >> 6, off = q & 63u;           // group of 64 items, offset in the group
      if (cur == CUR_IDLE && rank < avail) {
        // item q -> local tile q / (64 spp), position in the 8x8 tile (q / spp) % 64, sample q % spp
        const uint32_t lt = small_div ? fastdiv24(g2, (uint32_t)fr.spp, rcp_spp) : g2 / (uint32_t)fr.spp;
        const uint32_t rem = (g2 - lt * (uint32_t)fr.spp) * 64u + off;          // < 64 spp
        const uint32_t pos = small_div ? fastdiv24(rem, (uint32_t)fr.spp, rcp_spp) : rem / (uint32_t)fr.spp;
        const uint32_t smp = rem - pos * (uint32_t)fr.spp;
        const uint32_t gt = (uint32_t)fr.tile_begin + lt;  // index into this rank's tile list
        const uint32_t tile = fr.tiles ? (uint32_t)fr.tiles[gt] : gt;
        const uint32_t trow = small_div ? fastdiv24(tile, (uint32_t)fr.tiles_x, rcp_tx) : tile / (uint32_t)fr.tiles_x;
        const int px = (int)(tile - trow * (uint32_t)fr.tiles_x) * 8 + (int)(pos & 7u);
        const int py = (int)trow * 8 + (int)(pos >> 3);
        if (px < fr.W && py < fr.H) {                      // items outside the image are consumed and dropped
          pixel = (uint32_t)(py * fr.W + px);
          item = q;
          if (smp == 0u && fr.out_primid) fr.out_primid[fr.packed ? gt * 64u + pos : pixel] = -1;
          camera_ray(fr, px, py, pixel, smp, ro, rd);
          if (fr.maxb > 0) {
            rec_st(lrec, LR_ATT, mk3(1, 1, 1)); rec_st(lrec, LR_FA, mk3(0, 0, 0));
            rec_st(lrec, LR_FL, mk3(-inf, -inf, -inf)); rec_st(lrec, LR_FH, mk3(inf, inf, inf));
            rec_sti(lrec, LR_SB, (int)smp, 0);
            if (RR) rec[PR_S * 64] = 1.0;
            li = -1;
            RT_BEGIN_QUERY(RT_TMAX);
          } else {
            // traceLoop _ 0: the sample is the background along its camera ray (Trace.hs:22); nothing is traced
            const d3 b_ = background(sc, rd);
            double *o_ = fr.samples + 3 * (size_t)q;
            o_[0] = b_.x; o_[1] = b_.y; o_[2] = b_.z;
          }
        }
      }
      pool_next += n_need < avail ? n_need : avail;
    }
  };

#ifdef RT_TIMERS          // experiment build: per-body wall time in the TIMED kernel variant (no counters, no extra spills)
  constexpr bool TIMERS = true;
#else
  constexpr bool TIMERS = STATS;
#endif
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = TIMERS ? clock64() : 0ull;   // vote, node, prim, shade, refill; 5..11: SHADE's sub-blocks (RT_TIMERS)
#define RT_TICK(k) do { if (TIMERS) { const unsigned long long n_ = clock64(); tacc[k] += n_ - tlast; tlast = n_; } } while (0)
#ifdef RT_TIMERS
#define RT_TICK_SH(k) RT_TICK(k)
#else
#define RT_TICK_SH(k) do { } while (0)
#endif
// census of SHADE's sub-blocks (STATS variant): lanes that ran block k, and trips in which any lane did
#define RT_SH_COUNT(k, cond) do { if (STATS) { const unsigned long long m_ = __ballot(cond); if (cond) { st[STAT_SH_L0 + (k)]++; if (lane == __ffsll((long long)m_) - 1) st[STAT_SH_W0 + (k)]++; } } } while (0)
  for (;;) {
    {
      const unsigned long long m_fin = __ballot(cur == CUR_FIN || cur == CUR_FIN0);
      const int n_wait = __popcll(m_fin | __ballot(cur == CUR_IDLE));           // all 64 idle at the start
      // finish + refill once enough lanes wait for it (when nothing else is left in the wave all 64 do, so the last paths are never stranded)
      static_assert(RT_REFILL_MIN <= 64, "a wave whose lanes all wait must pass the threshold");
      if (n_wait >= RT_REFILL_MIN && (m_fin != 0ull || !exhausted)) {
        RT_TICK(0);
        RT_MARK("FINISH_BEGIN");
        if (m_fin != 0ull) finish_paths();
        RT_MARK("REFILL_BEGIN");
        if (!exhausted) refill();
        RT_MARK("REFILL_DONE");
        RT_TICK(4);
      }
    }
    const bool cur_leaf = (uint32_t)cur > (uint32_t)CUR_SPECIAL_MAX;
    const bool at_node = cur >= 0 || cur == CUR_POP || (SPEC && cur_leaf && pend == 0 && sp > 0);
    const bool at_leaf = cur_leaf || (SPEC && pend != 0);
    const bool is_done = cur == CUR_DONE && !(SPEC && pend != 0);
    const int n_node = __popcll(__ballot(at_node));
    const int n_leaf = __popcll(__ballot(at_leaf));
    const int n_done = __popcll(__ballot(is_done));
    const int n_act = SPEC ? __popcll(__ballot(cur != CUR_IDLE && cur != CUR_FIN && cur != CUR_FIN0)) : n_node + n_leaf + n_done;
    if (n_act == 0) break;                                 // nothing in flight (every waiting lane was finished above) and the refill found nothing: frame done
    if (STATS && lane == 0) st[STAT_LIVE] += (uint32_t)n_act;

    // The traversal bodies and SHADE are two guarded blocks in a row, not one if / else-if / else chain.  The compiler linearises a
    // chain of uniform branches around divergent code into blocks joined by "flow" merges whose skipped side is undefined, and the
    // ray, its box constants and the shading state (redefined by SHADE only) then lived in two register sets, reconciled by ~37
    // register copies at the end of EVERY trip of this loop, NODE and PRIM trips included (found in the ISA: tools/isa_counts.py).
    // A guarded block merges its results with the values that reach it, in place.  `trip` is made opaque between the blocks so
    // that they are not threaded back into a chain.  (NODE, PRIM and SHADE as three guarded blocks removes the last copies too
    // but costs 6-8 more spilled VGPRs, with reloads in the PRIM body: 47.0 ms against 43.0 for this form and 44.2 for the chain.)
    int trip = (n_done * RT_SHADE_DEN >= n_act * RT_SHADE_NUM || n_node + n_leaf == 0) ? 2
               : (n_node * RT_NODE_WEIGHT >= n_leaf * RT_LEAF_WEIGHT ? 0 : 1);
    trip = __builtin_amdgcn_readfirstlane(trip);
    asm volatile("" : "+s"(trip));
    if (trip == 0) {
      // ================================================= NODE step (BVH.hs:123-143)
      if (STATS) st[STAT_IT_NODE]++;
      RT_TICK(0);
      RT_MARK("NODE_BEGIN");
      RT_PRIO(RT_PRIO_NODE);
#pragma unroll
      for (int it = 0; it < NODE_UNROLL; ++it) {
        bool pp = cur == CUR_POP;                            // lanes that still owe a pop from the previous trip
        if (SPEC) {                                          // postpone a freshly reached leaf and go on with the next stack entry
          const bool spec = (uint32_t)cur > (uint32_t)CUR_SPECIAL_MAX && pend == 0 && sp > 0;
          pend = spec ? cur : pend;
          pp = pp || spec;
        }
        const bool isn = cur >= 0;
        if (STATS && isn) st[STAT_NODES]++;
        bool hL, hR;
        BT tL, tR;
        int refL = 0, refR = 0;
        if constexpr (sizeof(BT) == 4) {
          // Lanes that are not at a node read node 0 (always present; a broadcast) instead of being masked off:
          // no zero-initialised registers, no exec-mask branch around the loads; their results are discarded by `isn`.
          float4 q0, q1, q2;
          int2 q3;
          const int ni = isn ? cur : 0;
          if (RT_TOP_NODES_MAX > 0 && ni < top_nodes_v) {   // top of the tree: the workgroup's LDS copy (ds_read_b128)
            const float4 *p = lnodes + 4 * ni;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = *(const int2 *)(p + 3);
            asm volatile("" ::: "memory");                   // keep this an LDS load: do not merge with the global path into a FLAT load
            if (STATS && isn) st[STAT_TOPHIT]++;
          } else {                                           // scalar base + 32-bit lane offset
            const char *p = (const char *)nodes + (size_t)((uint32_t)ni << 6);
            q0 = *(const float4 *)p; q1 = *(const float4 *)(p + 16); q2 = *(const float4 *)(p + 32); q3 = *(const int2 *)(p + 48);
          }
          const float bxL[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
          const float bxR[6] = {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
          hL = box_test(br, bxL, tminb, tmaxb, tL) && isn;
          hR = box_test(br, bxR, tminb, tmaxb, tR) && isn;
          refL = q3.x; refR = q3.y;
        } else {
          const DNode<BT> nd = nodes[isn ? cur : 0];
          hL = box_test(br, nd.box, tminb, tmaxb, tL) && isn;
          hR = box_test(br, nd.box + 6, tminb, tmaxb, tR) && isn;
          refL = nd.ref[0]; refR = nd.ref[1];
        }
        const bool lf = tL < tR;                             // BVH.hs:132: nearer child first, tie => right first
        if (hL && hR) {                                      // push the farther child with its entry distance
          const BT ts = lf ? tR : tL;
          float tsf;
          if constexpr (sizeof(BT) == 8) tsf = __double2float_rd((double)ts); else tsf = (float)ts;
          const uint2 ent = make_uint2((uint32_t)(lf ? refR : refL), __float_as_uint(tsf));
          if (sp < RT_STACK_LDS) stack[sp * 64] = ent; else ovf[(sp - RT_STACK_LDS) * 64] = ent;
          if constexpr (sizeof(BT) == 8) xst[sp * 64] = (double)ts;
          ++sp;
        }
        cur = (hL || hR) ? ((hL && (lf || !hR)) ? refL : refR) : cur;
        pp = pp || (isn && !(hL || hR));
        RT_POP2(pp);
      }
      RT_PRIO(0);
      RT_TICK(1);
    } else if (trip == 1) {
      // ================================================= PRIM step (closestHitList, HitHelpers.hs:11-16)
      RT_MARK("NODE_END");
      if (STATS) st[STAT_IT_PRIM]++;
      RT_TICK(0);
      RT_MARK("PRIM_BEGIN");
      RT_PRIO(RT_PRIO_PRIM);
#pragma unroll
      for (int it = 0; it < PRIM_UNROLL; ++it) {
        const bool use_p = SPEC && pend != 0;               // the postponed leaf comes first: traversal order
        const bool isl = use_p || (uint32_t)cur > (uint32_t)CUR_SPECIAL_MAX;
        const int pi = ~(use_p ? pend : cur);
        // lanes not in a leaf read primitive 0 (always present) and discard the result: see the NODE body
        const DPrim P = *(const DPrim *)((const char *)sc.tprims + (size_t)((uint32_t)(isl ? pi : 0) * 80u));
        const uint32_t kind = P.kind & DPRIM_KIND_MASK;
        // Moller-Trumbore without early exits (Triangle.hs:30-53): same operations, same order, the rejections become one predicate
        double t;
        bool h, lit_done = false;
        {
          const d3 p0 = mk3(P.d[0], P.d[1], P.d[2]), e1 = mk3(P.d[3], P.d[4], P.d[5]), e2 = mk3(P.d[6], P.d[7], P.d[8]);
          const d3 hh = cross3(rd, e2);                      // :34
          const double a = dot3(e1, hh);                     // :35
          const double f = 1.0 / a;                          // :39
          const d3 sv = ro - p0;                             // :40
          const double u = f * dot3(sv, hh);                 // :41
          const d3 q = cross3(sv, e1);                       // :45
          const double v = f * dot3(rd, q);                  // :46
          t = f * dot3(e2, q);                               // :50
          h = !(fabs(a) < 1e-8) && !(u < 0.0 || u > 1.0) && !(v < 0.0 || u + v > 1.0) && (t >= RT_TMIN && t <= best_t);   // :36,42,47,53
        }
        if (STATS && isl && kind == RT_PRIM_TRIANGLE) st[STAT_TRI]++;
        if (__ballot(isl && kind != RT_PRIM_TRIANGLE) != 0ull) {   // wave-uniform: leaves with spheres/planes are rare
          RT_MARK("PRIM_NONTRI_BEGIN");
          // The ray enters this branch through an opaque copy: otherwise the compiler hoists what depends on the ray alone
          // (the plane's three FP64 reciprocals 1/d and its parallel-axis tests, the sphere's d.d: ~55 of ~190 VALU
          // instructions) in front of the branch and every triangle test pays for it.
          d3 ro_ = ro, rd_ = rd;
          asm volatile("" : "+v"(ro_.x), "+v"(ro_.y), "+v"(ro_.z), "+v"(rd_.x), "+v"(rd_.y), "+v"(rd_.z));
          int ps = PLANE_MISS;
          if (isl && kind == RT_PRIM_SPHERE) {
            if (STATS) st[STAT_SPHERE]++;
            h = hit_sphere(P, ro_, rd_, RT_TMIN, best_t, t);
            // A sphere of negative radius lies outside its own (inside-out) box c -+ r (Sphere.hs:20-22), and that box is not
            // inside its ancestors' boxes either (a union by min / max does not contain the hull of an inside-out box); a
            // sphere of radius 0 is a point on the faces of its own box, where every hit is rounding noise.  Both are seen only
            // through the interval the reference's traversal hands their leaf (the chain of clipped ancestor intervals), so
            // a candidate hit settles the whole query with the literal walk.  d[6] < 0: the leaf is the root (raw interval).
            // A sphere of POSITIVE radius touches its exact box (no pad, unlike Triangle.hs:24-28): where it is hit at such a point
            // t1 can fall an ulp outside the interval its leaf receives (Sphere.hs:26), so a candidate hit is classified against
            // that interval like a plane's.
            // Only a hit within rounding of a box face can fall outside: a hit point at least r / 1000 inside the sphere's own box on
            // every axis (a box inside the leaf's, and rounding is monotone) is inside the leaf's slab interval by r / (1000 |d|),
            // which is compared with a bound of 1e-12 x (the magnitudes that enter the slab distances and t1) - four orders above their
            // rounding errors; such hits skip the exact re-test of the leaf's box (3 FP64 divisions), the others take it.
            if (h) {
              if (!(P.d[3] > 0.0)) { if (P.d[6] >= 0.0) { ps = PLANE_LITERAL; h = false; } }
              else if (P.d[6] >= 0.0) {
                const double r_ = P.d[3], m_ = 0.999 * r_;
                const double qx = fabs((ro_.x + t * rd_.x) - P.d[0]), qy = fabs((ro_.y + t * rd_.y) - P.d[1]), qz = fabs((ro_.z + t * rd_.z) - P.d[2]);
                const double mag = (fabs(ro_.x) + fabs(ro_.y) + fabs(ro_.z)) + (fabs(P.d[0]) + fabs(P.d[1]) + fabs(P.d[2])) + (r_ + t * (fabs(rd_.x) + fabs(rd_.y) + fabs(rd_.z)));
                // (an axis under the parallel rule, |d| < 1e-8, is decided by the ORIGIN's position, BoundingBox.hs:21-22: exact path)
                const bool deep = qx <= m_ && qy <= m_ && qz <= m_ && 0.001 * r_ > 1e-12 * mag &&
                                  !(fabs(rd_.x) < 1e-7) && !(fabs(rd_.y) < 1e-7) && !(fabs(rd_.z) < 1e-7);
                if (!deep) { ps = classify_in_leaf((long long)P.d[6], sc.plane_boxes, ro_, rd_, RT_TMIN, best_t, t); h = ps == PLANE_HIT; }
              }
            }
          } else if (isl && kind == RT_PRIM_PLANE) {
            if (STATS) st[STAT_PLANE]++;
            ps = hit_plane(P, sc.plane_boxes, ro_, rd_, RT_TMIN, best_t, t);
            h = ps == PLANE_HIT;
          }
          {
            if (ps == PLANE_LITERAL) {
              // The plane is crossed beyond its leaf box's exit: the reference sees it only if that leaf is visited second
              // and its sibling produced a hit (BVH.hs:137-138) - a property of the reference's traversal, so the whole
              // query is answered by the literal walk of the reference's tree and this lane's traversal ends here.
              RT_COLD_ARGS();
              if (STATS) st[STAT_LITERAL]++;
              // (box_precision = 1 does not stop a shadow query at its first hit: one that already holds a hit is blocked whatever
              // follows.  The FP32-box kernels never get here with a hit - and this form keeps 9 VGPR spills out of their PRIM body.)
              bool ask = true;
              if constexpr (sizeof(BT) == 8) ask = !(li >= 0 && best >= 0);
              const LitHit lh = literal_hit((const DNode<double> *)sc.ref_nodes64, sc.prims, sc.ref_root_ref, rec + (size_t)fr.lit_row * 64,
                                            ro_.x, ro_.y, ro_.z, rd_.x, rd_.y, rd_.z, RT_TMIN, li >= 0 ? best_t : RT_TMAX);
              if (ask) { best = lh.best; best_t = lh.t; tmaxb = tmax_for_boxes<BT>(lh.t); }
              lit_done = true;
            }
          }
        }
        RT_MARK("PRIM_NONTRI_END");
        h = h && isl;
        // FAST: this tree's order is not the reference's.  FP32 boxes on the reference's tree: two children's entry distances
        // can compare differently after rounding, so the visiting order is not always the reference's either.  Both settle
        // exact ties (duplicated objects) the reference's way; the FP64 boxes visit in its order and need no rule.
        if constexpr (FAST || sizeof(BT) == 4) {
          const bool tie = h && li < 0 && best >= 0 && t == best_t;
          if (__ballot(tie) != 0ull) {
            RT_MARK("PRIM_TIE_BEGIN");
            if (tie) h = tie_candidate_wins(sc.leaf_anc, sc.ref_anc, sc.ref_depth, (const DNode<double> *)sc.ref_nodes64,
                                            ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, FAST ? (int)P.mat : pi, best);
            RT_MARK("PRIM_TIE_END");
          }
        }
        best_t = h ? t : best_t;                             // t <= best_t: a later element wins ties
        best = h ? (FAST ? P.mat : pi) : best;               // FAST: the slot holds the primitive's index in the reference's order
        tmaxb = h ? tmax_for_boxes<BT>(t) : tmaxb;
        // shadow ray: any hit blocks (Light.hs:30-32 only tests Just / Nothing), so the query stops at the first one - except in
        // box_precision = 1, the literal walk, which like the reference runs every shadow query as a full closest-hit query
        const bool blocked = h && li >= 0 && sizeof(BT) == 4;
        const bool last = (P.kind & DPRIM_LAST) != 0u;
        if (SPEC) pend = use_p ? ((blocked || last) ? 0 : pend - 1) : (blocked ? 0 : pend);
        cur = isl ? (blocked ? CUR_DONE : ((last || use_p) ? cur : cur - 1)) : cur;   // cur - 1 == ~(pi + 1)
        bool pp = isl && !use_p && !blocked && last;
        RT_POP2(pp);
        cur = lit_done ? CUR_DONE : cur;                     // (rare) the query was answered by literal_hit
        sp = lit_done ? 0 : sp;
        if (SPEC) pend = lit_done ? 0 : pend;
      }
      RT_PRIO(0);
      RT_TICK(2);
    }
    asm volatile("" : "+s"(trip));
    if (trip == 2) {
      // ================================================= SHADE step: lanes whose query finished
      if (STATS) st[STAT_IT_SHADE]++;
      RT_TICK(0);
      RT_MARK("SHADE_BEGIN");
      RT_PRIO(RT_PRIO_SHADE);
      if (is_done) {                                         // (every sub-block re-reads the arguments it needs: RT_COLD_ARGS)
        double *r_ = rec;
        asm volatile("" : "+v"(r_) : : "memory");           // the record is re-read from memory, never carried in registers
        gdouble *r = (gdouble *)r_;
        const bool anyhit = li >= 0;
        const bool is_h = !anyhit && best >= 0;              // closest-hit query found a surface
        bool need_scatter = false, need_finish = false;
        double q_tmax = RT_TMAX;
        bool need_ray0 = false;                              // background along the PRIMARY ray (depth exhausted, Trace.hs:22)
        // ---- All memory this trip can need is requested up front, in two rounds of independent loads, instead of one
        // dependent round trip per use: SHADE is latency bound (records and scene compete for 4 MB of L2 per XCD).
        // Round 1: the whole record and the header of the hit primitive.
        int2 sb = rec_ldi(lrec, LR_SB);
        d3 fa = rec_ld(lrec, LR_FA), att = rec_ld(lrec, LR_ATT), nrm = mk3(0, 0, 0);
        d3 fl = rec_ld(lrec, LR_FL), fh = rec_ld(lrec, LR_FH);
        int2 pkm;                                            // (material, kind) of the hit primitive
        d3 tn, tn2;                                          // triangles: face normal n (Triangle.hs:51) and normalize(n) (Light.hs:19)
        DMaterial M;
        {
          RT_COLD_ARGS();
          const size_t pidx = (size_t)(uint32_t)(is_h ? best : (anyhit ? (hitp & ~HITP_FRONT) : 0));   // primitive just hit / of the pending hit
          const double *sr_ = sc.shade_recs + 8 * pidx;       // one 64-byte line
          pkm = *(const int2 *)(sr_ + 6);
          tn = ld3(sr_);
          tn2 = ld3(sr_ + 3);
          // Round 2: the material (a small, cache-resident table); the face-forwarded normal of a triangle hit is
          // not kept in the record, it is +-n of the remembered primitive.
          if (anyhit) {
            if (((uint32_t)pkm.y & DPRIM_KIND_MASK) == RT_PRIM_TRIANGLE) nrm = (hitp & HITP_FRONT) ? tn : neg3(tn);
            else nrm = rec_ld(r, PR_NRM);
          }
          M = sc.mats[(is_h || anyhit) ? pkm.x : 0];
        }
        RT_MARK("SH_LOADS_END");
        RT_TICK_SH(5);
        const bool hit_tri = ((uint32_t)pkm.y & DPRIM_KIND_MASK) == RT_PRIM_TRIANGLE;
        RT_SH_COUNT(0, !anyhit && best >= 0);
        RT_SH_COUNT(1, anyhit && best < 0);
        if (!anyhit) {
          if (best < 0) {                                    // Trace.hs:25: background along the current ray
            need_finish = true;
          } else {
            RT_COLD_ARGS();
            if (sb.x == 0 && sb.y == 0 && fr.out_primid)   // primary hit of sample 0: the pixel's id (slot of a packed output: tile*64 + position)
              fr.out_primid[fr.packed ? (uint32_t)fr.tile_begin * 64u + item / (uint32_t)fr.spp : pixel] = sc.prim_ids[best];
            const uint32_t kind = (uint32_t)pkm.y & DPRIM_KIND_MASK;
            const d3 hp = ro + best_t * rd;                  // R.at ray t
            d3 outward;
            if (kind == RT_PRIM_TRIANGLE) outward = tn;      // normalize(e1 x e2), Triangle.hs:51, precomputed at upload
            else {
              const DPrim P = sc.prims[best];
              if (kind == RT_PRIM_SPHERE) outward = (1.0 / P.d[3]) * (hp - mk3(P.d[0], P.d[1], P.d[2]));   // Sphere.hs:38
              else outward = mk3(P.d[3], P.d[4], P.d[5]);                                                   // Plane.hs:29
            }
            const bool front = dot3(rd, outward) < 0;        // Class.hs:24
            nrm = front ? outward : neg3(outward);           // :25
            if (kind != RT_PRIM_TRIANGLE) rec_st(r, PR_NRM, nrm);
            if (M.flags & (DMAT_GLASS_DIR | DMAT_METAL))    // unitDir (Trace.hs:32) is only used by the glass and metal bounces
              rec_st(r, PR_UDIR, normalize3(rd));
            hitp = best | (front ? HITP_FRONT : 0);
            if (RR) fa = fa + r[PR_S * 64] * ld3(M.emission);   // (extension) weighted by the roulette survivals so far
            else fa = fa + ld3(M.emission);                  // emitted (Trace.hs:29) joins the pending offset
            rec_st(lrec, LR_FA, fa);
            ro = hp;                                         // shadow rays and the bounce ray start at the un-offset hit point
            if (STATS) st[STAT_SHADED]++;
            if (sc.n_lights > 0) li = sc.n_lights - 1; else need_scatter = true;
          }
        }
        RT_MARK("SH_HIT_END");
        RT_TICK_SH(6);
        if (anyhit) {
          if (best < 0) {                                    // unblocked: Light.hs:36-38 / :48-49
            RT_COLD_ARGS();
            const DLight L = sc.lights[li];
            d3 lnormal;                                      // normalize (hitNormal hit), Light.hs:19
            if (hit_tri) lnormal = (hitp & HITP_FRONT) ? tn2 : neg3(tn2);   // precomputed; normalize(-n) == -normalize(n) exactly
            else lnormal = normalize3(nrm);
            const double lightPower = fmax(0.0, dot3(lnormal, rd));
            d3 contrib;
            if (L.kind == RT_LIGHT_POINT) {
              const d3 tl = ld3(L.vec) - ro;
              const double dist = sqrt(dot3(tl, tl));
              const double attenuation = 1 / (dist * dist);
              contrib = (lightPower * attenuation) * ld3(L.intensity);
            } else {
              contrib = lightPower * ld3(L.intensity);
            }
            if (RR) fa = fa + r[PR_S * 64] * (ld3(M.diffuse) * contrib);
            else fa = fa + ld3(M.diffuse) * contrib;         // litColor = surfaceColor * directLight (Trace.hs:36), light by light
            rec_st(lrec, LR_FA, fa);
          }
          --li;
          if (li < 0) need_scatter = true;
        }
        RT_MARK("SH_LIGHT_END");
        RT_TICK_SH(7);
        RT_SH_COUNT(2, !need_scatter && !need_finish);
        if (!need_scatter && !need_finish) {                 // shadow ray towards light li (Light.hs:26-30, 39-42)
          RT_COLD_ARGS();
          // FAST: a light the surface faces away from adds max(0, n.l) * ... = 0 whether or not it is visible
          // (Light.hs:36-38, 48-49), so its shadow query is not traced; the next light, or the bounce, follows in this trip.
          d3 lnormal = mk3(0, 0, 0);
          if (FAST) lnormal = hit_tri ? ((hitp & HITP_FRONT) ? tn2 : neg3(tn2)) : normalize3(nrm);   // as in the light term above
          for (;;) {
            const DLight L = sc.lights[li];
            if (L.kind == RT_LIGHT_POINT) {
              const d3 tl = ld3(L.vec) - ro;
              rd = normalize3(tl);
              q_tmax = sqrt(dot3(tl, tl)) - 0.01;
            } else {
              rd = normalize3(-1.0 * ld3(L.vec));
              q_tmax = RT_TMAX;
            }
            if (!FAST || dot3(lnormal, rd) > 0.0) break;
            if (STATS) st[STAT_SHADOW_CULLED]++;
            if (--li < 0) { need_scatter = true; q_tmax = RT_TMAX; break; }   // the bounce ray follows in this trip
          }
        }
        RT_MARK("SH_SHADOWSETUP_END");
        RT_TICK_SH(8);
        RT_SH_COUNT(3, need_scatter);
        if (need_scatter) {                                  // Trace.hs:27-76
          RT_COLD_ARGS();
          const bool front = (hitp & HITP_FRONT) != 0;
          if (RR && !(M.flags & DMAT_NO_ATTEN)) { att = att * ld3(M.diffuse); rec_st(lrec, LR_ATT, att); }   // (the roulette needs the new attenuation first)
          // (extension, off unless rt_params.rr_start_bounce > 0) Russian roulette before the next bounce is traced: survive with
          // probability q = clamp(max channel of the new attenuation, 0.05, 1); a survivor's deeper value is weighted 1/q outside
          // the clamp, a terminated path adds nothing deeper: F(x) = clamp(s x + a, L, H) with s the product of the weights so far
          double ten = 10.0;
          bool rr_dead = false;
          if (RR) {
            double sw = r[PR_S * 64];
            const int nb = sb.y + 1;
            if (nb >= fr.rr_start && nb < fr.maxb) {
              const double q = fmin(1.0, fmax(0.05, fmax(att.x, fmax(att.y, att.z))));
              const u4 w2 = philox4x32_10(pixel, (uint32_t)sb.x, 1u + (uint32_t)sb.y, 1u, fr.key0, fr.key1);
              rr_dead = !(u01(w2.x) < q);
              sw = sw * (1.0 / q);
            }
            if (rr_dead) { att = mk3(0, 0, 0); rec_st(lrec, LR_ATT, att); }      // the sample is F(0): finished below without a fold
            else { r[PR_S * 64] = sw; ten = 10.0 * sw; }
          }
          // fold `c + clamp(., 0, 10)` into the pending outer function F(x) = clamp(x + a, L, H); a already holds a + c:
          //   L' = clamp(a, L, H), H' = clamp(a + 10, L, H)      (DESIGN.md §Iterative form of the nested clamp)
          if (!rr_dead) {
            const d3 nl = clamp3_finite(fa, fl, fh), nh = clamp3_finite(mk3(fa.x + ten, fa.y + ten, fa.z + ten), fl, fh);
            fl = nl; fh = nh;
            rec_st(lrec, LR_FL, fl);
            rec_st(lrec, LR_FH, fh);
          }
          const u4 w = philox4x32_10(pixel, (uint32_t)sb.x, 1u + (uint32_t)sb.y, 0u, fr.key0, fr.key1);
          const d3 randVec = normalize3(mk3(um11(w.x), um11(w.y), um11(w.z)));   // :38-42
          const double randD = u01(w.w);                                         // :41
          if (M.flags & DMAT_GLASS_DIR) {                                        // :48-59
            const d3 udir = rec_ld(r, PR_UDIR);
            const double eta = front ? 1.0 / M.ior : M.ior;
            const double cosTheta = fmin(-dot3(udir, nrm), 1.0);
            const double sinTheta = sqrt(1.0 - cosTheta * cosTheta);
            const double r0 = (1 - eta) / (1 + eta);                             // schlick, :85-89
            const double r0sq = r0 * r0;
            const double reflectProb = r0sq + (1 - r0sq) * pow5_cr(1 - cosTheta);   // (1-cos)**5 correctly rounded = libm's pow but for 1 x in 1,000 (rt_pow5.h)
            if (eta * sinTheta > 1.0 || randD < reflectProb) rd = reflect3(udir, nrm);
            else rd = refract3(udir, nrm, eta);
          } else if (M.flags & DMAT_METAL) {                                     // :60-64
            rd = normalize3(reflect3(rec_ld(r, PR_UDIR), nrm) + 0.05 * randVec);
          } else {                                                               // :65-67
            rd = normalize3(nrm + randVec);
          }
          if (!RR && !(M.flags & DMAT_NO_ATTEN)) { att = att * ld3(M.diffuse); rec_st(lrec, LR_ATT, att); }   // :69-72
          sb.y += 1;
          rec_sti(lrec, LR_SB, sb.x, sb.y);
          li = -1;
          if (sb.y >= fr.maxb) { need_ray0 = true; need_finish = true; }         // depth 0: background along ray0, :22
          if (rr_dead) { need_finish = true; rec_sti(lrec, LR_SB, sb.x, -1); }   // (extension) the sample is clamp(a, L, H): finish_paths adds no background
        }
        RT_MARK("SH_SCATTER_END");
        RT_TICK_SH(9);
        RT_SH_COUNT(4, need_finish);
        if (need_finish) cur = need_ray0 ? CUR_FIN0 : CUR_FIN;   // path end: the sample is written in the next refill step (finish_paths)
        RT_MARK("SH_FINISH_END");
        RT_TICK_SH(10);
        if (!need_finish) { RT_COLD_ARGS(); RT_BEGIN_QUERY(q_tmax); }
        RT_TICK_SH(11);
      }
      RT_PRIO(0);
      RT_MARK("SHADE_END");
      RT_TICK(3);
      RT_MARK("REFILL_END");
    }
  }
#undef RT_BEGIN_QUERY
#undef RT_POP2

  if (TIMERS && lane == 0) {
    for (int k = 0; k < 12; ++k) atomicAdd(fr.stats + STAT_T_VOTE + k, tacc[k]);
  }
#ifdef RT_TIMERS
  if (fr.dbg && lane == 0) {                               // per wave: start, pool exhausted, end (100 MHz wall clock), chunks taken
    unsigned long long *d_ = fr.dbg + 4 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    d_[0] = dbg_t0; d_[1] = dbg_texh; d_[2] = wall_clock64(); d_[3] = dbg_chunks;
  }
#endif
  if (STATS) {
#pragma unroll
    for (int i = 0; i < STAT_T_VOTE; ++i) {
      unsigned long long v = wave_sum((unsigned long long)st[i]);
      if (lane == 0 && v) atomicAdd(fr.stats + i, v);
    }
  }
}
