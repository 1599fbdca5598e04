// rt_kernel_v2.h — the render kernel as a VOTE-SCHEDULED per-lane state machine.
//
// v1 ran one complete BVH query per lane per outer trip, so every lane waited for the longest
// traversal of its wave (measured VALU lane utilisation 22 %, profiles/r01_summary.md).  Here every
// lane carries its traversal state (current ref, stack pointer, closest hit) across trips and each
// trip of the wave executes exactly ONE of three bodies, picked by wave-uniform ballots:
//     NODE  step : lanes at an internal node fetch it, test both child boxes, descend / push / pop
//     PRIM  step : lanes in a leaf test one primitive and advance / pop
//     SHADE step : lanes whose query finished run the path state transition (light loop, material
//                  bounce, path end, next sample) and start their next query
// so a finished lane never idles until the slowest ray of the wave is done, and the expensive bodies
// run with as many lanes as possible.  Semantics are those of v1 (same results bit for bit: the
// per-lane sequence of operations is unchanged, only its interleaving across lanes differs).
#pragma once

#ifndef RT_NODE_WEIGHT
#define RT_NODE_WEIGHT 1    // NODE step runs while n_node * WEIGHT >= n_leaf
#endif
#ifndef RT_SHADE_NUM
#define RT_SHADE_NUM 1     // run the SHADE body once >= NUM/DEN of the wave's live lanes wait for it
#define RT_SHADE_DEN 4
#endif

template <typename BT, bool STATS>
__global__ __launch_bounds__(256, RT_MIN_WAVES) void k_render_tiles(const KScene sc, const KFrame fr) {
  extern __shared__ uint2 lds_stack[];                     // [wave][entry][lane]
  const int lane = threadIdx.x & 63;
  uint2 *stack = lds_stack + (size_t)(threadIdx.x >> 6) * fr.stack_depth * 64 + lane;
  const DNode<BT> *nodes = (const DNode<BT> *)sc.nodes;
  const double inf = __builtin_huge_val();
  const BT tminb = tmin_for_boxes<BT>();

  uint32_t st[STAT_COUNT];
#pragma unroll
  for (int i = 0; i < STAT_COUNT; ++i) st[i] = 0;

  for (;;) {
    uint32_t ti = 0;
    if (lane == 0) ti = atomicAdd(fr.tile_counter, 1u);
    ti = __builtin_amdgcn_readfirstlane(ti);
    if (ti >= (uint32_t)fr.n_tiles) break;
    const int tile = fr.tiles ? fr.tiles[ti] : (int)ti;
    const int px = (tile % fr.tiles_x) * 8 + (lane & 7);
    const int py = (tile / fr.tiles_x) * 8 + (lane >> 3);
    const bool inimg = px < fr.W && py < fr.H;
    const uint32_t pixel = (uint32_t)(py * fr.W + px);

    d3 sum = mk3(0, 0, 0);
    int pid0 = -1;

    if (inimg && fr.maxb <= 0) {
      // traceLoop _ 0: every sample is the background along its camera ray (Trace.hs:22)
      for (int s = 0; s < fr.spp; ++s) {
        d3 o, d;
        camera_ray(fr, px, py, pixel, (uint32_t)s, o, d);
        sum = sum + background(sc, d);
      }
    }

    // ---- path state
    d3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1), udir = mk3(0, 0, 0), nrm = mk3(0, 0, 0), direct = mk3(0, 0, 0);
    d3 atten = mk3(1, 1, 1), Fa = mk3(0, 0, 0), FL = mk3(-inf, -inf, -inf), FH = mk3(inf, inf, inf);
    int s = 0, bounce = 0, li = -1, hit_mat = 0;
    bool front = false;
    double q_tmax = RT_TMAX;
    // ---- query (traversal) state
    BoxRay<BT> br;
    BT tmaxb = (BT)0;
    double best_t = 0.0;
    int best = -1, cur = 0, sp = 0;
    bool ray_done = false;
    bool active = inimg && fr.maxb > 0;

#define RT_BEGIN_QUERY()                                                           \
  do {                                                                             \
    make_boxray(br, ro, rd, sc.rscene);                                            \
    best_t = q_tmax; tmaxb = tmax_for_boxes<BT>(best_t);                           \
    best = -1; sp = 0; cur = sc.root_ref; ray_done = false;                        \
    if (STATS) st[li >= 0 ? STAT_SHADOW : STAT_CLOSEST]++;                         \
  } while (0)
#define RT_POP()                                                                   \
  do {                                                                             \
    for (;;) {                                                                     \
      if (sp == 0) { ray_done = true; break; }                                     \
      --sp;                                                                        \
      const uint2 e_ = stack[sp * 64];                                             \
      if (!cull_entry(__uint_as_float(e_.y), tmaxb)) { cur = (int)e_.x; break; }   \
    }                                                                              \
  } while (0)

    if (active) {
      camera_ray(fr, px, py, pixel, 0u, ro, rd);
      RT_BEGIN_QUERY();
    }

    for (;;) {
      const unsigned long long m_act = __ballot(active);
      if (m_act == 0ull) break;
      const bool at_node = active && !ray_done && cur >= 0;
      const bool at_leaf = active && !ray_done && cur < 0;
      const int n_act = __popcll(m_act);
      const int n_node = __popcll(__ballot(at_node));
      const int n_leaf = __popcll(__ballot(at_leaf));
      const int n_done = n_act - n_node - n_leaf;

      if (n_done * RT_SHADE_DEN >= n_act * RT_SHADE_NUM || n_node + n_leaf == 0) {
        // ================================================= SHADE step: lanes whose query finished
        if (active && ray_done) {
          const bool anyhit = li >= 0;
          bool need_scatter = false, need_finish = false;
          d3 T = mk3(0, 0, 0);
          if (!anyhit) {
            if (best < 0) {                                    // Trace.hs:25
              T = atten * background(sc, rd);
              need_finish = true;
            } else {
              if (s == 0 && bounce == 0) pid0 = sc.prim_ids[best];
              const DPrim P = sc.prims[best];
              const uint32_t kind = P.kind & DPRIM_KIND_MASK;
              hit_mat = P.mat;
              const d3 hp = ro + best_t * rd;                  // R.at ray t
              d3 outward;
              if (kind == RT_PRIM_TRIANGLE) outward = normalize3(cross3(mk3(P.d[3], P.d[4], P.d[5]), mk3(P.d[6], P.d[7], P.d[8])));   // Triangle.hs:51
              else if (kind == RT_PRIM_SPHERE) outward = (1.0 / P.d[3]) * (hp - mk3(P.d[0], P.d[1], P.d[2]));                        // Sphere.hs:38
              else outward = mk3(P.d[3], P.d[4], P.d[5]);                                                                            // Plane.hs:29
              front = dot3(rd, outward) < 0;                   // Class.hs:24
              nrm = front ? outward : neg3(outward);           // :25
              udir = normalize3(rd);                           // Trace.hs:32
              ro = hp;                                         // shadow rays and the bounce ray start at the un-offset hit point
              direct = mk3(0, 0, 0);
              if (STATS) st[STAT_SHADED]++;
              if (sc.n_lights > 0) li = sc.n_lights - 1; else need_scatter = true;
            }
          } else {
            if (best < 0) {                                    // unblocked: Light.hs:36-38 / :48-49
              const DLight L = sc.lights[li];
              const d3 lnormal = normalize3(nrm);              // Light.hs:19
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
              direct = contrib + direct;                       // foldr V.add, Light.hs:21
            }
            --li;
            if (li < 0) need_scatter = true;
          }
          if (!need_scatter && !need_finish) {                 // shadow ray towards light li (Light.hs:26-30, 39-42)
            const DLight L = sc.lights[li];
            if (L.kind == RT_LIGHT_POINT) {
              const d3 tl = ld3(L.vec) - ro;
              rd = normalize3(tl);
              q_tmax = sqrt(dot3(tl, tl)) - 0.01;
            } else {
              rd = normalize3(-1.0 * ld3(L.vec));
              q_tmax = RT_TMAX;
            }
          }
          if (need_scatter) {                                  // Trace.hs:27-76
            const DMaterial M = sc.mats[hit_mat];
            const d3 surf = ld3(M.diffuse);
            const d3 c = (ld3(M.emission) + surf * direct);    // emitted + litColor
            // fold `c + clamp(., 0, 10)` into the pending outer function F(x) = clamp(x + Fa, FL, FH)
            const d3 ca = c + Fa;
            const d3 nL = clamp3(ca, FL, FH);
            const d3 nH = clamp3(mk3(ca.x + 10.0, ca.y + 10.0, ca.z + 10.0), FL, FH);
            FL = nL; FH = nH; Fa = ca;
            const u4 w = philox4x32_10(pixel, (uint32_t)s, 1u + (uint32_t)bounce, 0u, fr.key0, fr.key1);
            const d3 randVec = normalize3(mk3(um11(w.x), um11(w.y), um11(w.z)));   // :38-42
            const double randD = u01(w.w);                                         // :41
            if (M.flags & DMAT_GLASS_DIR) {                                        // :48-59
              const double eta = front ? 1.0 / M.ior : M.ior;
              const double cosTheta = fmin(-dot3(udir, nrm), 1.0);
              const double sinTheta = sqrt(1.0 - cosTheta * cosTheta);
              const double r0 = (1 - eta) / (1 + eta);                             // schlick, :85-89
              const double r0sq = r0 * r0;
              const double reflectProb = r0sq + (1 - r0sq) * pow(1 - cosTheta, 5.0);
              if (eta * sinTheta > 1.0 || randD < reflectProb) rd = reflect3(udir, nrm);
              else rd = refract3(udir, nrm, eta);
            } else if (M.flags & DMAT_METAL) {                                     // :60-64
              rd = normalize3(reflect3(udir, nrm) + 0.05 * randVec);
            } else {                                                               // :65-67
              rd = normalize3(nrm + randVec);
            }
            if (!(M.flags & DMAT_NO_ATTEN)) atten = atten * surf;                  // :69-72
            ++bounce;
            li = -1;
            q_tmax = RT_TMAX;
            if (bounce >= fr.maxb) {                                               // depth 0: background along ray0, :22
              d3 o0, d0;
              camera_ray(fr, px, py, pixel, (uint32_t)s, o0, d0);
              T = atten * background(sc, d0);
              need_finish = true;
            }
          }
          if (need_finish) {
            sum = sum + clamp3(T + Fa, FL, FH);
            ++s;
            if (s < fr.spp) {
              camera_ray(fr, px, py, pixel, (uint32_t)s, ro, rd);
              atten = mk3(1, 1, 1); Fa = mk3(0, 0, 0); FL = mk3(-inf, -inf, -inf); FH = mk3(inf, inf, inf);
              bounce = 0; li = -1; q_tmax = RT_TMAX;
            } else {
              active = false;
            }
          }
          if (active) RT_BEGIN_QUERY();
        }
      } else if (n_node * RT_NODE_WEIGHT >= n_leaf) {
        // ================================================= NODE step (BVH.hs:123-143)
        if (at_node) {
          const DNode<BT> nd = nodes[cur];
          if (STATS) st[STAT_NODES]++;
          BT tL, tR;
          const bool hL = box_test(br, nd.box, tminb, tmaxb, tL);
          const bool hR = box_test(br, nd.box + 6, tminb, tmaxb, tR);
          if (hL && hR) {
            const bool lf = tL < tR;                           // :132 (tie => right first)
            const BT ts = lf ? tR : tL;
            float tsf;
            if constexpr (sizeof(BT) == 8) tsf = __double2float_rd((double)ts); else tsf = (float)ts;
            stack[sp * 64] = make_uint2((uint32_t)(lf ? nd.ref[1] : nd.ref[0]), __float_as_uint(tsf));
            ++sp;
            cur = lf ? nd.ref[0] : nd.ref[1];
          } else if (hL) {
            cur = nd.ref[0];
          } else if (hR) {
            cur = nd.ref[1];
          } else {
            RT_POP();
          }
        }
      } else {
        // ================================================= PRIM step (closestHitList, HitHelpers.hs:11-16)
        if (at_leaf) {
          const int pi = ~cur;
          const DPrim P = sc.prims[pi];
          const uint32_t kind = P.kind & DPRIM_KIND_MASK;
          double t;
          bool h;
          if (kind == RT_PRIM_TRIANGLE) { if (STATS) st[STAT_TRI]++; h = hit_triangle(P, ro, rd, RT_TMIN, best_t, t); }
          else if (kind == RT_PRIM_SPHERE) { if (STATS) st[STAT_SPHERE]++; h = hit_sphere(P, ro, rd, RT_TMIN, best_t, t); }
          else { if (STATS) st[STAT_PLANE]++; h = hit_plane(P, sc.plane_boxes, ro, rd, RT_TMIN, best_t, t); }
          if (h) {                                             // t <= best_t: a later element wins ties
            best_t = t; best = pi;
            tmaxb = tmax_for_boxes<BT>(best_t);
            if (li >= 0) ray_done = true;                      // shadow ray: any hit blocks (Light.hs:30-32)
          }
          if (!ray_done) {
            if (P.kind & DPRIM_LAST) RT_POP(); else cur = cur - 1;   // ~(pi+1)
          }
        }
      }
    }
#undef RT_BEGIN_QUERY
#undef RT_POP

    if (inimg) {
      const d3 avg = (1.0 / (double)fr.spp) * sum;                               // RenderWorker.hs:114
      const double e = fr.exposure;
      const d3 toned = mk3(avg.x * e / (avg.x * e + 1), avg.y * e / (avg.y * e + 1), avg.z * e / (avg.z * e + 1));   // Trace.hs:96-101
      const d3 col = mk3(pow(toned.x, fr.gamma), pow(toned.y, fr.gamma), pow(toned.z, fr.gamma));                    // :103-105
      const size_t slot = fr.packed ? (size_t)ti * 64 + lane : (size_t)py * fr.W + px;
      if (fr.out_color) { fr.out_color[3 * slot] = col.x; fr.out_color[3 * slot + 1] = col.y; fr.out_color[3 * slot + 2] = col.z; }
      if (fr.out_radiance) { fr.out_radiance[3 * slot] = avg.x; fr.out_radiance[3 * slot + 1] = avg.y; fr.out_radiance[3 * slot + 2] = avg.z; }
      if (fr.out_primid) fr.out_primid[slot] = pid0;
    }
  }

  if (STATS) {
#pragma unroll
    for (int i = 0; i < STAT_COUNT; ++i) {
      unsigned long long v = wave_sum((unsigned long long)st[i]);
      if (lane == 0 && v) atomicAdd(fr.stats + i, v);
    }
  }
}
