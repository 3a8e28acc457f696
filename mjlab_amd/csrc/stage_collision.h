// stage_collision.h -- stage 2: static pair list + box terrain, analytic primitives.
// Part of kernels.h (included there, in this order, by every translation unit of the library); not a
// stand-alone header.
#pragma once

// ====================================================================================
// Stage 2: collision (static candidate pair list; plane/sphere/capsule/box primitives)
// ====================================================================================
struct RawCon { float dist, pos[3], frame[6]; };
// dst = take ? src : dst, field by field (v_cndmask).  Contact slots are filled through VALUE
// selects with compile-time slot indices: a conditional store to `slot[n]` makes the compiler keep
// the whole slot array in scratch memory.
__device__ __forceinline__ void rc_take(RawCon& dst, const RawCon& src, bool take) {
  dst.dist = take ? src.dist : dst.dist;
#pragma unroll
  for (int k = 0; k < 3; ++k) dst.pos[k] = take ? src.pos[k] : dst.pos[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) dst.frame[k] = take ? src.frame[k] : dst.frame[k];
}

__device__ __forceinline__ int plane_sphere(RawCon* c, float margin, const float* ppos, const float* pn, const float* spos, float r) {
  float dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  float cdist = dot3(dif, pn);
  if (cdist > margin + r) return 0;
  c->dist = cdist - r;
  for (int k = 0; k < 3; ++k) { c->pos[k] = spos[k] + pn[k] * (-c->dist * 0.5f - r); c->frame[k] = pn[k]; c->frame[3 + k] = 0.f; }
  return 1;
}
__device__ __forceinline__ int sphere_sphere(RawCon* c, float margin, const float* p1, float r1, const float* p2, float r2) {
  float dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  float cd2 = dot3(dif, dif), mn = margin + r1 + r2;
  if (cd2 > mn * mn) return 0;
  float len = sqrtf(cd2);
  if (len < MINVAL) { dif[0] = 1.f; dif[1] = dif[2] = 0.f; }
  else { float inv = 1.0f / len; dif[0] *= inv; dif[1] *= inv; dif[2] *= inv; }
  c->dist = len - r1 - r2;
  for (int k = 0; k < 3; ++k) { c->pos[k] = p1[k] + dif[k] * (r1 + c->dist * 0.5f); c->frame[k] = dif[k]; c->frame[3 + k] = 0.f; }
  return 1;
}
__device__ __forceinline__ int capsule_capsule(RawCon* c, float margin, const float* pos1, const float* axis1, const float* size1,
                               const float* pos2, const float* axis2, const float* size2) {
  float dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  float ma = dot3(axis1, axis1), mb = -dot3(axis1, axis2), mc = dot3(axis2, axis2);
  float u = -dot3(axis1, dif), v = dot3(axis2, dif), det = ma * mc - mb * mb;
  float vec1[3], vec2[3];
  if (fabsf(det) >= MINVAL) {
    float x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > size1[1]) { x1 = size1[1]; x2 = (v - mb * size1[1]) / mc; }
    else if (x1 < -size1[1]) { x1 = -size1[1]; x2 = (v + mb * size1[1]) / mc; }
    if (x2 > size2[1]) { x2 = size2[1]; x1 = clipf((u - mb * size2[1]) / ma, -size1[1], size1[1]); }
    else if (x2 < -size2[1]) { x2 = -size2[1]; x1 = clipf((u + mb * size2[1]) / ma, -size1[1], size1[1]); }
    for (int k = 0; k < 3; ++k) { vec1[k] = pos1[k] + axis1[k] * x1; vec2[k] = pos2[k] + axis2[k] * x2; }
    return sphere_sphere(c, margin, vec1, size1[0], vec2, size2[0]);
  }
  // parallel axes: up to two contacts out of four end-point candidates, taken in order.  The
  // output slots are written with compile-time indices (a run-time `c + n` would push the
  // whole contact array into scratch memory).
  int n = 0;
  RawCon t;
  auto push = [&](bool ok) {
    rc_take(c[0], t, ok && n == 0);
    rc_take(c[1], t, ok && n == 1);
    n += ok ? 1 : 0;
  };
  float x1, x2;
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] + axis1[k] * size1[1];
  x2 = clipf((v - mb * size1[1]) / mc, -size2[1], size2[1]);
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] + axis2[k] * x2;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] - axis1[k] * size1[1];
  x2 = clipf((v + mb * size1[1]) / mc, -size2[1], size2[1]);
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] + axis2[k] * x2;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  if (n == 2) return n;
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] + axis2[k] * size2[1];
  x1 = clipf((u - mb * size2[1]) / ma, -size1[1], size1[1]);
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] + axis1[k] * x1;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  if (n == 2) return n;
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] - axis2[k] * size2[1];
  x1 = clipf((u + mb * size2[1]) / ma, -size1[1], size1[1]);
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] + axis1[k] * x1;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  return n < 2 ? n : 2;
}


// Sphere vs (static) box: centre into the box frame, clamp; outside the normal runs along
// clamped point -> centre, inside through the nearest face.  Normal points from the sphere
// (geom1) into the box (geom2), pos midway between the surfaces.  bmat is row major (world =
// bmat * local).
__device__ __forceinline__ int sphere_box(RawCon* c, float margin, const float* spos, float r, const float* bpos, const float* bmat, const float* bsize) {
  const float dif[3] = {spos[0] - bpos[0], spos[1] - bpos[1], spos[2] - bpos[2]};
  float loc[3], dv[3], nl[3], pl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) loc[i] = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) dv[i] = loc[i] - clipf(loc[i], -bsize[i], bsize[i]);
  const float d2 = dot3(dv, dv), mn = margin + r;
  if (d2 > mn * mn) return 0;
  if (d2 > 0.f) {
    const float len = sqrtf(d2);
    c->dist = len - r;
#pragma unroll
    for (int i = 0; i < 3; ++i) { nl[i] = dv[i] / len; pl[i] = (loc[i] - dv[i]) + nl[i] * (c->dist * 0.5f); }
  } else {
    // centre inside the box: leave through the nearest face (first one on ties)
    int k = 0;
    float depth = bsize[0] - fabsf(loc[0]);
#pragma unroll
    for (int i = 1; i < 3; ++i) { const float di = bsize[i] - fabsf(loc[i]); if (di < depth) { depth = di; k = i; } }
#pragma unroll
    for (int i = 0; i < 3; ++i) nl[i] = (i == k) ? (loc[i] >= 0.f ? 1.f : -1.f) : 0.f;
    c->dist = -depth - r;
#pragma unroll
    for (int i = 0; i < 3; ++i) pl[i] = loc[i] + nl[i] * ((depth - r) * 0.5f);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c->pos[i] = bpos[i] + bmat[3 * i] * pl[0] + bmat[3 * i + 1] * pl[1] + bmat[3 * i + 2] * pl[2];
    c->frame[i] = -(bmat[3 * i] * nl[0] + bmat[3 * i + 1] * nl[1] + bmat[3 * i + 2] * nl[2]);
    c->frame[3 + i] = 0.f;
  }
  return 1;
}
// d/dt of half the squared distance between pc + t h (box frame) and the box, and the squared distance
__device__ __forceinline__ float seg_box_slope(const float* pc, const float* h, const float* bsize, float t) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { const float p = pc[i] + t * h[i]; s += (p - clipf(p, -bsize[i], bsize[i])) * h[i]; }
  return s;
}
__device__ __forceinline__ float seg_box_dist2(const float* pc, const float* h, const float* bsize, float t) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { const float p = pc[i] + t * h[i], e = p - clipf(p, -bsize[i], bsize[i]); s += e * e; }
  return s;
}
// Capsule vs box: up to 4 sphere_box() contacts of spheres of the capsule's radius on its axis
// (point cpos + axis * halflen * t): the two ends, plus the ends ta <= tb of the interval where
// the axis is closest to the box when they are interior points outside the box (if the axis runs
// through the box with both ends outside: the inside point nearest to the capsule's centre).  The distance
// along the axis is convex, so its slope is monotone: ta / tb come from two bisections.
#define MJLAB_CAPBOX_ITERS 24
__device__ __forceinline__ int capsule_box(RawCon* c, float margin, const float* cpos, const float* axis, const float* csize, const float* bpos,
                                           const float* bmat, const float* bsize) {
  const float dif[3] = {cpos[0] - bpos[0], cpos[1] - bpos[1], cpos[2] - bpos[2]};
  float pc[3], h[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    pc[i] = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
    h[i] = (bmat[i] * axis[0] + bmat[3 + i] * axis[1] + bmat[6 + i] * axis[2]) * csize[1];
  }
  const float sm = seg_box_slope(pc, h, bsize, -1.f), sp = seg_box_slope(pc, h, bsize, 1.f);
  // ta = smallest t with slope >= 0, tb = largest t with slope <= 0; both searches run in one loop
  float alo = -1.f, ahi = 1.f, blo = -1.f, bhi = 1.f;
  for (int it = 0; it < MJLAB_CAPBOX_ITERS; ++it) {
    const float am = 0.5f * (alo + ahi), bm = 0.5f * (blo + bhi);
    const bool ag = seg_box_slope(pc, h, bsize, am) >= 0.f, bl = seg_box_slope(pc, h, bsize, bm) <= 0.f;
    ahi = ag ? am : ahi; alo = ag ? alo : am;
    blo = bl ? bm : blo; bhi = bl ? bhi : bm;
  }
  const float ta = sm >= 0.f ? -1.f : (sp < 0.f ? 1.f : ahi);
  const float tb = sp <= 0.f ? 1.f : (sm > 0.f ? -1.f : blo);
  const float eps = 1e-6f;
  const bool ia = ta > -1.f + eps && ta < 1.f - eps, ib = tb > -1.f + eps && tb < 1.f - eps;
  const bool oa = seg_box_dist2(pc, h, bsize, ta) > 0.f, ob = seg_box_dist2(pc, h, bsize, tb) > 0.f;
  // the axis runs THROUGH the box with both ends outside (a thin capsule across an edge, deeper than
  // its radius): the inside point nearest to the capsule's centre carries the contact (the middle of
  // a chord through opposite faces would be equidistant from both)
  const bool pierce = ia && ib && !oa && !ob;
  const float tmid = pierce ? clipf(0.f, ta, tb) : ta;
  const bool use_a = pierce || (ia && oa);
  const bool use_b = ib && tb - ta > eps && ob;
  int n = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float t = q == 0 ? 1.f : (q == 1 ? -1.f : (q == 2 ? tmid : tb));
    const bool use = q < 2 ? true : (q == 2 ? use_a : use_b);
    float p[3];
    RawCon tc;
    for (int k = 0; k < 3; ++k) p[k] = cpos[k] + axis[k] * (csize[1] * t);
    const bool hit = use && sphere_box(&tc, margin, p, csize[0], bpos, bmat, bsize) != 0;
    for (int k = 0; k < 3; ++k) tc.frame[3 + k] = axis[k];
    rc_take(c[0], tc, hit && n == 0); rc_take(c[1], tc, hit && n == 1); rc_take(c[2], tc, hit && n == 2); rc_take(c[3], tc, hit && n == 3);
    n += hit ? 1 : 0;
  }
  return n;
}

// Moving box (geom1) vs a static terrain box (geom2): candidate `cand` of this repository's documented
// rule (NOT mjc_BoxBox): 0..7 corners of the moving box as points against the
// terrain box (on a face exactly the plane-box contacts), 8..15 corners of the terrain box against the moving
// box (normal flipped), 16..39 the terrain box's 12 edges clipped to the inside of the moving box (slab
// clipping; the points at 1/4 and 3/4 of the inside interval, each leaving through the moving box's nearest
// face).  The first 4 hits in candidate order are the pair's contacts; the normal points into the terrain.
#define MJLAB_BOXBOX_NCAND 40
__device__ __forceinline__ int box_box_candidate(RawCon* c, int cand, float margin, const float* pos, const float* mat, const float* size,
                                                 const float* bpos, const float* bmat, const float* bsize) {
  if (cand < 8) {
    float vec[3], corner[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) vec[k] = ((cand >> k) & 1) ? size[k] : -size[k];
    mul_mat_vec3(corner, mat, vec);
    for (int k = 0; k < 3; ++k) corner[k] += pos[k];
    return sphere_box(c, margin, corner, 0.f, bpos, bmat, bsize);
  }
  float pt[3];
  if (cand < 16) {
    const int i = cand - 8;
    float vec[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) vec[k] = ((i >> k) & 1) ? bsize[k] : -bsize[k];
    mul_mat_vec3(pt, bmat, vec);
    for (int k = 0; k < 3; ++k) pt[k] += bpos[k];
  } else {
    const int e = (cand - 16) >> 1, smp = (cand - 16) & 1, a = e >> 2;
    float v0[3], v1[3], e0[3], e1[3], q0[3], q1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // axis a runs -size..+size; the other two sit at the extremes picked by the bits of e
      const int rel = (k - a + 3) % 3;  // 0: the edge's own axis, 1 / 2: the other two (bit 0 / bit 1 of e)
      const float ext = ((e >> (rel == 2 ? 1 : 0)) & 1) ? bsize[k] : -bsize[k];  // rel == 0: unused (no shift by -1)
      v0[k] = rel == 0 ? -bsize[k] : ext;
      v1[k] = rel == 0 ? bsize[k] : ext;
    }
    mul_mat_vec3(e0, bmat, v0);
    mul_mat_vec3(e1, bmat, v1);
    for (int k = 0; k < 3; ++k) { e0[k] += bpos[k] - pos[k]; e1[k] += bpos[k] - pos[k]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      q0[i] = mat[i] * e0[0] + mat[3 + i] * e0[1] + mat[6 + i] * e0[2];
      q1[i] = mat[i] * e1[0] + mat[3 + i] * e1[1] + mat[6 + i] * e1[2];
    }
    float t0 = 0.f, t1 = 1.f;
    bool miss = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float h = q1[i] - q0[i];
      if (fabsf(h) < MINVAL) { miss |= fabsf(q0[i]) > size[i]; continue; }
      float ta = (-size[i] - q0[i]) / h, tb = (size[i] - q0[i]) / h;
      if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
      t0 = fmaxf(t0, ta);
      t1 = fminf(t1, tb);
    }
    if (miss || t1 - t0 <= 1e-6f) return 0;
    const float t = t0 + (t1 - t0) * (smp ? 0.75f : 0.25f);
    for (int k = 0; k < 3; ++k) pt[k] = pos[k] + e0[k] + t * (e1[k] - e0[k]);
  }
  if (!sphere_box(c, margin, pt, 0.f, pos, mat, size)) return 0;
  for (int k = 0; k < 3; ++k) c->frame[k] = -c->frame[k];
  return 1;
}

// Terrain broadphase of one moving geom: walk the grid cells under its bounding sphere and keep
// the (at most MJLAB_TCAND_MAX, smallest ids first) boxes within reach in ascending order in
// `cand` (this lane's LDS slots).  A box listed in several cells is looked at once, in the lowest
// cell the two footprints share.
// `lc` = the geom's centre in the world's local frame (world = lc + org): grid cells and box tops are looked up in world
// coordinates, distances are formed in the local frame.
__device__ __forceinline__ int terrain_walk(const Model& m, const float* lc, const float (&org)[3], float reach, int* cand, bool* dropped) {
  const int nx = m.size.tgrid_nx, ny = m.size.tgrid_ny;
  const float x0 = (float)m.opt.tgrid_x0, y0 = (float)m.opt.tgrid_y0, inv = 1.0f / (float)m.opt.tgrid_cell;
  const float centre[3] = {lc[0] + org[0], lc[1] + org[1], lc[2] + org[2]};
  int ix0 = (int)floorf((centre[0] - reach - x0) * inv), ix1 = (int)floorf((centre[0] + reach - x0) * inv);
  int iy0 = (int)floorf((centre[1] - reach - y0) * inv), iy1 = (int)floorf((centre[1] + reach - y0) * inv);
  ix0 = min(max(ix0, 0), nx - 1); ix1 = min(max(ix1, 0), nx - 1);
  iy0 = min(max(iy0, 0), ny - 1); iy1 = min(max(iy1, 0), ny - 1);
  int n = 0;
  for (int ix = ix0; ix <= ix1; ++ix)
    for (int iy = iy0; iy <= iy1; ++iy) {
      const int c = ix * ny + iy;
      if (centre[2] - reach > m.tgrid_ztop[c]) continue;  // wholly above everything in this cell
      const int kend = m.tgrid_start[c + 1];
      for (int k = m.tgrid_start[c]; k < kend; ++k) {
        const int b = m.tgrid_item[k];
        const int bx = m.tbox_cell0[2 * b], by = m.tbox_cell0[2 * b + 1];
        if (ix != max(ix0, bx) || iy != max(iy0, by)) continue;
        const float *bpos = m.tbox_pos + 3 * b, *bmat = m.tbox_mat + 9 * b, *bsize = m.tbox_size + 3 * b;
        const float dif[3] = {lc[0] - (bpos[0] - org[0]), lc[1] - (bpos[1] - org[1]), lc[2] - (bpos[2] - org[2])};
        float d2 = 0.f;
        for (int i = 0; i < 3; ++i) {
          const float loc = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
          const float dv = loc - clipf(loc, -bsize[i], bsize[i]);
          d2 += dv * dv;
        }
        if (d2 > reach * reach) continue;
        // sorted insert, bounded: the largest id falls off the end
        int pos = n;
        while (pos > 0 && cand[pos - 1] > b) --pos;
        *dropped |= n >= MJLAB_TCAND_MAX;  // this box or the largest id in the list falls off
        if (pos >= MJLAB_TCAND_MAX) continue;
        for (int q = n < MJLAB_TCAND_MAX ? n : MJLAB_TCAND_MAX - 1; q > pos; --q) cand[q] = cand[q - 1];
        cand[pos] = b;
        n = n < MJLAB_TCAND_MAX ? n + 1 : n;
      }
    }
  return n;
}

__device__ __forceinline__ void make_frame(float* f9, const float* f6) {
  float x[3] = {f6[0], f6[1], f6[2]}, y[3] = {f6[3], f6[4], f6[5]};
  if (sqrtf(dot3(y, y)) < 0.5f) {
    y[0] = y[1] = y[2] = 0.f;
    if (x[1] < 0.5f && x[1] > -0.5f) y[1] = 1.f; else y[2] = 1.f;
  }
  float t = dot3(x, y);
  y[0] -= t * x[0]; y[1] -= t * x[1]; y[2] -= t * x[2];
  normalize3(y);
  float z[3];
  cross3(z, x, y);
  for (int k = 0; k < 3; ++k) { f9[k] = x[k]; f9[3 + k] = y[k]; f9[6 + k] = z[k]; }
}

// LDS: poses (12) and constants (8: type, size, rbound, margin, gap) of geoms [geom_lds0, ngeom) | per moving geom TCAND_MAX
// candidate boxes | flat pair lists (non-box, box)
__host__ __device__ inline int collision_lds_floats(const mjlab_sizes_t& s) {
  const int lists = 3 * s.ntgeom * MJLAB_TCAND_MAX;  // terrain lists; the close-pair list (npair) aliases them
  return 20 * (s.ngeom - s.geom_lds0) + (lists > s.npair ? lists : s.npair);
}

// Contact parameters (mj_contactParam) of the pair (g1, g2) + ordered append of this lane's n
// raw contacts at slots base + off ..
__device__ __forceinline__ void emit_contacts(const Model& m, const Data& d, int w, int g1, int g2, float margin, float gap, const RawCon (&rc)[4],
                                              int n, int first, const float* gfri, const float* gsolref, const float* gsolimp, const float* gsolmix,
                                              const float (&org)[3]) {
  const int ncm = m.size.nconmax;
  int condim;
  float fri[3], solref[2], solimp[5];
  const int pr1 = m.geom_priority[g1], pr2 = m.geom_priority[g2];
  if (pr1 != pr2) {
    const int gi = pr1 > pr2 ? g1 : g2;
    condim = m.geom_condim[gi];
    for (int k = 0; k < 3; ++k) fri[k] = fmaxf(gfri[3 * gi + k], MINMU);  // (mj_contactParam: friction never below mjMINMU; the cone rows divide by it)
    for (int k = 0; k < 2; ++k) solref[k] = gsolref[2 * gi + k];
    for (int k = 0; k < 5; ++k) solimp[k] = gsolimp[5 * gi + k];
  } else {
    condim = max(m.geom_condim[g1], m.geom_condim[g2]);
    for (int k = 0; k < 3; ++k) fri[k] = fmaxf(fmaxf(gfri[3 * g1 + k], gfri[3 * g2 + k]), MINMU);  // (one v_max3_f32)
    const float sm1 = gsolmix[g1], sm2 = gsolmix[g2];
    float mix;
    if (sm1 >= MINVAL && sm2 >= MINVAL) mix = sm1 / (sm1 + sm2);
    else if (sm1 < MINVAL && sm2 < MINVAL) mix = 0.5f;
    else if (sm1 < MINVAL) mix = 0.f;
    else mix = 1.f;
    if (gsolref[2 * g1] > 0.f && gsolref[2 * g2] > 0.f)
      for (int k = 0; k < 2; ++k) solref[k] = mix * gsolref[2 * g1 + k] + (1.f - mix) * gsolref[2 * g2 + k];
    else
      for (int k = 0; k < 2; ++k) solref[k] = fminf(gsolref[2 * g1 + k], gsolref[2 * g2 + k]);
    for (int k = 0; k < 5; ++k) solimp[k] = mix * gsolimp[5 * g1 + k] + (1.f - mix) * gsolimp[5 * g2 + k];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // no early exit: rc[i] must stay a compile-time index (registers, not scratch)
    const int c = first + i;
    if (i >= n || c >= ncm) continue;
    float f9[9];
    make_frame(f9, rc[i].frame);
    const size_t wc = (size_t)w * ncm + c;
    d.contact_dist[wc] = rc[i].dist;
    for (int k = 0; k < 3; ++k) { d.contact_pos[3 * wc + k] = rc[i].pos[k] + org[k]; d.contact_prel[3 * wc + k] = rc[i].pos[k]; }
    for (int k = 0; k < 9; ++k) d.contact_frame[9 * wc + k] = f9[k];
    d.contact_includemargin[wc] = margin - gap;
    float* f5 = d.contact_friction + 5 * wc;
    f5[0] = f5[1] = fri[0]; f5[2] = fri[1]; f5[3] = f5[4] = fri[2];
    for (int k = 0; k < 2; ++k) d.contact_solref[2 * wc + k] = solref[k];
    for (int k = 0; k < 5; ++k) d.contact_solimp[5 * wc + k] = solimp[k];
    d.contact_dim[wc] = condim;
    d.contact_geom[2 * wc] = g1;
    d.contact_geom[2 * wc + 1] = g2;
    d.contact_efc_address[wc] = -1;
  }
}

__device__ __forceinline__ void stage_collision(const Model& m, const Data& d, const int w, const int lane, const int flags, float* smem) {
  const int ng = m.size.ngeom, npair = m.size.npair;
  const int g0 = m.size.geom_lds0, nl = ng - g0;  // geoms [g0, ng) are staged; s_gx / s_gm are indexed by g - g0
  float* s_gx = smem;
  float* s_gm = s_gx + 3 * nl;
  PROF_INIT();
  float* s_gc = s_gm + 9 * nl;  // per staged geom: type (as int bits), size[3], rbound, margin, gap, -
  const float *gsize = MF(geom_size), *rbound = MF(geom_rbound), *gmargin = MF(geom_margin), *ggap = MF(geom_gap);
  const float *gfri = MF(geom_friction), *gsolref = MF(geom_solref), *gsolimp = MF(geom_solimp), *gsolmix = MF(geom_solmix);
  // The stage is bound by dependent global round trips: every per-geom constant the narrow phase
  // needs goes to LDS in this one batch, and the pair list is fetched one sweep ahead, so a sweep
  // finds all of its operands on chip.
  int ng1 = 0, ng2 = 0;  // geoms of pair (sweep 0, this lane)
  if (lane < npair) { ng1 = m.pair_geom[2 * lane]; ng2 = m.pair_geom[2 * lane + 1]; }
  // geom positions in the world's local frame: moving geoms as the position stage left them, static ones (posed once, in
  // world coordinates) minus the origin
  float org[3];
  for (int k = 0; k < 3; ++k) org[k] = d.xorigin[(size_t)w * 3 + k];
  {  // LDS-DMA (common.h): every lane names its source (static geom: world-frame array, moving geom: local-frame array); the
     // staged static geoms get the origin subtracted on chip after the barrier below -- the same subtraction, the same bits
    const int ng0 = m.size.nstaticgeom, ll = launder(lane);
    const float *gxw = d.geom_xpos + (size_t)w * ng * 3, *gxr = d.geom_xrel + (size_t)w * ng * 3;
    for (int k0 = 0; k0 < 3 * nl; k0 += 64) {
      const int e = 3 * g0 + k0 + ll;
      if (k0 + ll < 3 * nl) __builtin_amdgcn_global_load_lds((glds_src_t)((e < 3 * ng0 ? gxw : gxr) + e), (glds_dst_t)(s_gx + k0), 4, 0, 0);
    }
  }
  glds_to_lds(s_gm, d.geom_xmat + ((size_t)w * ng + g0) * 9, 9 * nl, lane);
  for (int l0 = lane; l0 < nl; l0 += 128) {  // two geoms per lane and trip: their 14 loads in flight together
    int gt[2];
    float gc[2][6];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int l = l0 + 64 * u, g = g0 + (l < nl ? l : 0);
      gt[u] = m.geom_type[g];
      for (int k = 0; k < 3; ++k) gc[u][k] = gsize[3 * g + k];
      gc[u][3] = rbound[g]; gc[u][4] = gmargin[g]; gc[u][5] = ggap[g];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int l = l0 + 64 * u;
      if (l < nl) {
        ((int*)s_gc)[8 * l] = gt[u];
        for (int k = 0; k < 6; ++k) s_gc[8 * l + 1 + k] = gc[u][k];
      }
    }
  }
  __syncthreads();
  {
    const int nst = 3 * (m.size.nstaticgeom - g0);  // staged static geoms (the plane of the flat scenes; none on the box terrains)
    if (nst > 0) {
      for (int k = lane; k < nst && k < 3 * nl; k += 64) { const int c = k - 3 * (k / 3); s_gx[k] = s_gx[k] - (c == 0 ? org[0] : c == 1 ? org[1] : org[2]); }
      __syncthreads();
    }
  }
  PROF_MARK(0);
  // ---- static pairs, pass 1: the cheap bounding test for every pair, survivors compacted IN PAIR
  // ORDER into an LDS list.  Few of the 502 G1 pairs are ever close, so the divergent narrow
  // phase below runs over one or two sweeps instead of eight.
  int* s_near = (int*)(s_gc + 8 * nl);  // (g1 << 16) | g2; aliases the terrain lists (built later)
  int nnear = 0;
  for (int p0 = 0; p0 < npair; p0 += 64) {
    const int p = p0 + lane;
    const int g1 = ng1, g2 = ng2;
    if (p + 64 < npair) { ng1 = m.pair_geom[2 * (p + 64)]; ng2 = m.pair_geom[2 * (p + 64) + 1]; }  // next sweep
    bool near = false;
    if (p < npair) {
      const int l1 = g1 - g0, l2 = g2 - g0;
      const float margin = fmaxf(s_gc[8 * l1 + 5], s_gc[8 * l2 + 5]);
      float dif[3];
      for (int k = 0; k < 3; ++k) dif[k] = s_gx[3 * l2 + k] - s_gx[3 * l1 + k];
      if (((const int*)s_gc)[8 * l1] == MJLAB_GEOM_PLANE) {
        const float z1[3] = {s_gm[9 * l1 + 2], s_gm[9 * l1 + 5], s_gm[9 * l1 + 8]};
        near = dot3(dif, z1) <= margin + s_gc[8 * l2 + 4];
      } else {
        const float bound = margin + s_gc[8 * l1 + 4] + s_gc[8 * l2 + 4];
        near = dot3(dif, dif) <= bound * bound;
      }
    }
    const unsigned long long nm = __ballot(near);
    if (near) s_near[nnear + __popcll(nm & ((1ull << lane) - 1ull))] = (int)(((unsigned)g1 << 16) | (unsigned)g2);
    nnear += __popcll(nm);
  }
  __syncthreads();
  int base = 0;  // contacts emitted so far (wave-uniform)
  // ---- pass 2: narrow phase over the close pairs
  for (int p0 = 0; p0 < nnear; p0 += 64) {
    const int p = p0 + lane;
    RawCon rc[4];
    int n = 0, g1 = 0, g2 = 0;
    float margin = 0.f, gap = 0.f;
    if (p < nnear) {
      const unsigned code = (unsigned)s_near[p];  // unsigned: geom ids up to 65535 (check_model)
      g1 = (int)(code >> 16); g2 = (int)(code & 0xffffu);
      const int l1 = g1 - g0, l2 = g2 - g0;
      const int t1 = ((const int*)s_gc)[8 * l1], t2 = ((const int*)s_gc)[8 * l2];
      margin = fmaxf(s_gc[8 * l1 + 5], s_gc[8 * l2 + 5]);
      gap = fmaxf(s_gc[8 * l1 + 6], s_gc[8 * l2 + 6]);
      float p1[3], p2[3], z1[3], z2[3], s1[3], s2[3];
      for (int k = 0; k < 3; ++k) {
        p1[k] = s_gx[3 * l1 + k]; p2[k] = s_gx[3 * l2 + k];
        z1[k] = s_gm[9 * l1 + 3 * k + 2]; z2[k] = s_gm[9 * l2 + 3 * k + 2];
        s1[k] = s_gc[8 * l1 + 1 + k]; s2[k] = s_gc[8 * l2 + 1 + k];
      }
      float dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      {
        if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_SPHERE) {
          n = plane_sphere(rc, margin, p1, z1, p2, s2[0]);
        } else if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_CAPSULE) {
          float q[3];
          RawCon t;
          for (int k = 0; k < 3; ++k) q[k] = p2[k] + z2[k] * s2[1];
          bool hit = plane_sphere(&t, margin, p1, z1, q, s2[0]) != 0;
          for (int k = 0; k < 3; ++k) t.frame[3 + k] = z2[k];
          rc_take(rc[0], t, hit);
          n = hit ? 1 : 0;
          for (int k = 0; k < 3; ++k) q[k] = p2[k] - z2[k] * s2[1];
          hit = plane_sphere(&t, margin, p1, z1, q, s2[0]) != 0;
          for (int k = 0; k < 3; ++k) t.frame[3 + k] = z2[k];
          rc_take(rc[0], t, hit && n == 0);
          rc_take(rc[1], t, hit && n == 1);
          n += hit ? 1 : 0;
        } else if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_BOX) {
          const float dist = dot3(dif, z1);
          float bm[9];
          for (int k = 0; k < 9; ++k) bm[k] = s_gm[9 * l2 + k];
          for (int i = 0; i < 8 && n < 4; ++i) {
            float vec[3] = {(i & 1) ? s2[0] : -s2[0], (i & 2) ? s2[1] : -s2[1], (i & 4) ? s2[2] : -s2[2]}, corner[3];
            mul_mat_vec3(corner, bm, vec);
            const float ldist = dot3(z1, corner);
            if (dist + ldist > margin || ldist > 0.f) continue;
            RawCon t;
            t.dist = dist + ldist;
            for (int k = 0; k < 3; ++k) {
              t.pos[k] = corner[k] + p2[k] + z1[k] * (-t.dist * 0.5f);
              t.frame[k] = z1[k]; t.frame[3 + k] = 0.f;
            }
            rc_take(rc[0], t, n == 0); rc_take(rc[1], t, n == 1); rc_take(rc[2], t, n == 2); rc_take(rc[3], t, n == 3);
            n++;
          }
        } else if (t1 == MJLAB_GEOM_SPHERE && t2 == MJLAB_GEOM_SPHERE) {
          n = sphere_sphere(rc, margin, p1, s1[0], p2, s2[0]);
        } else if (t1 == MJLAB_GEOM_SPHERE && t2 == MJLAB_GEOM_CAPSULE) {
          float vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
          const float x = clipf(dot3(z2, vec), -s2[1], s2[1]);
          for (int k = 0; k < 3; ++k) vec[k] = p2[k] + z2[k] * x;
          n = sphere_sphere(rc, margin, p1, s1[0], vec, s2[0]);
        } else if (t1 == MJLAB_GEOM_CAPSULE && t2 == MJLAB_GEOM_CAPSULE) {
          n = capsule_capsule(rc, margin, p1, z1, s1, p2, z2, s2);
        }
      }
    }
    int total;
    const int off = wave_excl_scan(n, lane, &total);
    if (n > 0) emit_contacts(m, d, w, g1, g2, margin, gap, rc, n, base + off, gfri, gsolref, gsolimp, gsolmix, org);
    base += total;
  }
  PROF_MARK(1);
  // ---- box terrain: moving spheres / capsules vs static boxes found through the xy grid ----
  const int ntg = m.size.ntgeom;
  bool tdrop = false;
  if (ntg > 0) {
    __syncthreads();  // the close-pair list shares its LDS with the lists built below
    int* s_cand = (int*)(s_gc + 8 * nl);            // [ntg][TCAND_MAX] box ids, ascending per geom
    int* s_pair = s_cand + ntg * MJLAB_TCAND_MAX;   // flat, ordered candidate list: (ti << 24) | slot
    int* s_pairb = s_pair + ntg * MJLAB_TCAND_MAX;  // the same for moving BOX geoms (own sweep below)
    int pbase = 0, bbase = 0;
    for (int t0 = 0; t0 < ntg; t0 += 64) {          // lanes = moving geoms
      const int ti = t0 + lane;
      int nc = 0;
      bool isbox = false;
      if (ti < ntg) {
        const int g = m.tgeom[ti];
        isbox = ((const int*)s_gc)[8 * (g - g0)] == MJLAB_GEOM_BOX;
        nc = terrain_walk(m, s_gx + 3 * (g - g0), org, s_gc[8 * (g - g0) + 4] + s_gc[8 * (g - g0) + 5], s_cand + ti * MJLAB_TCAND_MAX, &tdrop);
      }
      int total, totalb;
      const int off = wave_excl_scan(isbox ? 0 : nc, lane, &total);
      const int offb = wave_excl_scan(isbox ? nc : 0, lane, &totalb);
      int* dst = isbox ? s_pairb + bbase + offb : s_pair + pbase + off;
      for (int q = 0; q < nc; ++q) dst[q] = (ti << 24) | q;
      pbase += total;
      bbase += totalb;
    }
    __syncthreads();
    PROF_MARK(3);
    for (int p0 = 0; p0 < pbase; p0 += 64) {        // lanes = candidate (geom, box) pairs
      const int p = p0 + lane;
      RawCon rc[4];
      int n = 0, g = 0, gb = 0;
      float margin = 0.f, gap = 0.f;
      if (p < pbase) {
        const int code = s_pair[p], ti = code >> 24;
        const int b = s_cand[ti * MJLAB_TCAND_MAX + (code & 0xffffff)];
        g = m.tgeom[ti];
        gb = m.tbox_geom[b];
        const int l = g - g0;
        margin = s_gc[8 * l + 5];  // terrain boxes carry no margin / gap (checked when the model is compiled)
        gap = s_gc[8 * l + 6];
        float cp[3], cz[3], cs[3], bpos[3], bmat[9], bsize[3];
        for (int k = 0; k < 3; ++k) {
          cp[k] = s_gx[3 * l + k]; cz[k] = s_gm[9 * l + 3 * k + 2]; cs[k] = s_gc[8 * l + 1 + k];
          bpos[k] = m.tbox_pos[3 * b + k] - org[k]; bsize[k] = m.tbox_size[3 * b + k];
        }
        for (int k = 0; k < 9; ++k) bmat[k] = m.tbox_mat[9 * b + k];
        if (((const int*)s_gc)[8 * l] == MJLAB_GEOM_SPHERE) n = sphere_box(rc, margin, cp, cs[0], bpos, bmat, bsize);
        else n = capsule_box(rc, margin, cp, cz, cs, bpos, bmat, bsize);
      }
      int total;
      const int off = wave_excl_scan(n, lane, &total);
      if (n > 0) emit_contacts(m, d, w, g, gb, margin, gap, rc, n, base + off, gfri, gsolref, gsolimp, gsolmix, org);
      base += total;
    }
    // Moving boxes vs terrain boxes: one pair per sweep, lanes = the rule's 40 candidates (box_box_candidate);
    // the first 4 hits in candidate order are the pair's contacts.  Not mjc_BoxBox: see DESIGN.md section 7 (row 4).
    for (int p = 0; p < bbase; ++p) {
      RawCon rc[4];
      const int code = s_pairb[p], ti = code >> 24;
      const int b = s_cand[ti * MJLAB_TCAND_MAX + (code & 0xffffff)];
      const int g = m.tgeom[ti], gb = m.tbox_geom[b], l = g - g0;
      const float margin = s_gc[8 * l + 5], gap = s_gc[8 * l + 6];
      bool hit = false;
      if (lane < MJLAB_BOXBOX_NCAND) {
        float cp[3], cs[3], cm[9], bpos[3], bmat[9], bsize[3];
        for (int k = 0; k < 3; ++k) {
          cp[k] = s_gx[3 * l + k]; cs[k] = s_gc[8 * l + 1 + k];
          bpos[k] = m.tbox_pos[3 * b + k] - org[k]; bsize[k] = m.tbox_size[3 * b + k];
        }
        for (int k = 0; k < 9; ++k) { cm[k] = s_gm[9 * l + k]; bmat[k] = m.tbox_mat[9 * b + k]; }
        hit = box_box_candidate(rc, lane, margin, cp, cm, cs, bpos, bmat, bsize) != 0;
      }
      const unsigned long long hits = __ballot(hit);
      const int rank = __popcll(hits & ((1ull << lane) - 1ull));
      const int n = hit && rank < 4 ? 1 : 0;
      const int total = min(__popcll(hits), 4);
      if (n > 0) emit_contacts(m, d, w, g, gb, margin, gap, rc, n, base + rank, gfri, gsolref, gsolimp, gsolmix, org);
      base += total;
    }
  }
  const int ncm = m.size.nconmax;
  const bool anydrop = __ballot(tdrop) != 0ull;
  if (lane == 0) {
    d.ncon[w] = base < ncm ? base : ncm;
    d.overflow[w] = (base > ncm ? MJLAB_OVF_NCONMAX : 0) | (anydrop ? MJLAB_OVF_TCAND : 0);  // k_constraint adds MJLAB_OVF_NJMAX
  }
  PROF_MARK(2);
  PROF_FLUSH(d.profile + (size_t)w * 64 + 32);
}

#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64, 4) void k_collision(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  if ((flags & FLAG_FOLD) && d.fold_reuse[w]) return;
  stage_collision(m, d, w, lane, flags, smem);
}
#endif  // MJLAB_MAIN_TU
