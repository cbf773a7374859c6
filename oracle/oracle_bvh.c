/*
 * oracle_bvh.c -- line-by-line CPU restatement of the reference BLAS/TLAS
 * builder.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see jpt_oracle.h).
 * Citations: V = src/bvh/vec.h, H = src/bvh/bvh.h, C = src/bvh/bvh.cpp,
 * G = src/path_tracing/geometry_group3d.cpp, U = src/utils.h.
 *
 * Quirks reproduced on purpose:
 *  - BoundingBox() = {min (FLT_MAX,0,0,1), max (FLT_MIN,0,0,1)} because the
 *    one-argument vec4 ctor sets only x (C:6-10, V:49).
 *  - vec4 arithmetic touches w (V:53-71); centroid.w = 3 * 0.33333333f.
 *  - std::nth_element is restated from libstdc++ (introselect); the element
 *    order it leaves is libstdc++-specific.
 */
#include "jpt_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

typedef jpto_vec4 vec4;

/* std::min / std::max semantics (V:78-86): min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a */
static inline float smin(float a, float b) { return (b < a) ? b : a; }
static inline float smax(float a, float b) { return (a < b) ? b : a; }

static inline vec4 v4(float x, float y, float z, float w) { vec4 r = {x, y, z, w}; return r; }
static inline vec4 v4_add(vec4 a, vec4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline vec4 v4_sub(vec4 a, vec4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline vec4 v4_scale(vec4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline vec4 v4_min(vec4 a, vec4 b) { return v4(smin(a.x, b.x), smin(a.y, b.y), smin(a.z, b.z), smin(a.w, b.w)); }
static inline vec4 v4_max(vec4 a, vec4 b) { return v4(smax(a.x, b.x), smax(a.y, b.y), smax(a.z, b.z), smax(a.w, b.w)); }
static inline float v4_get(const vec4 *v, int i) /* V:94-126 */
{
    switch (i) { case 0: return v->x; case 1: return v->y; case 2: return v->z; case 3: return v->w; default: return v->x; }
}

typedef struct { vec4 min, max; } bbox_t;

/* C:6-10 */
static bbox_t bbox_default(void)
{
    bbox_t b;
    b.min = v4(FLT_MAX, 0.0f, 0.0f, 1.0f);
    b.max = v4(FLT_MIN, 0.0f, 0.0f, 1.0f);
    return b;
}
/* C:12-16 */
static void bbox_extend(bbox_t *b, vec4 p)
{
    b->min = v4_min(b->min, p);
    b->max = v4_max(b->max, p);
}
/* H:39-43 */
static float bbox_area(const bbox_t *b)
{
    vec4 d = v4_sub(b->max, b->min);
    return d.x * d.y + d.y * d.z + d.z * d.x;
}

struct jpto_builder {
    jpto_triangle *tris;       size_t n_tris, cap_tris;
    jpto_bvh_node *nodes;      size_t n_nodes, cap_nodes;
    jpto_blas_instance *inst;  size_t n_inst, cap_inst;
    jpto_tlas_node *tlas;      size_t n_tlas;
    jpto_tri_geometry *geom;
    jpto_tri_data *data;
};

static void *grow(void *p, size_t *cap, size_t need, size_t elem)
{
    if (need <= *cap) return p;
    size_t nc = *cap ? *cap * 2 : 64;
    while (nc < need) nc *= 2;
    *cap = nc;
    return realloc(p, nc * elem);
}

jpto_builder *jpto_builder_create(void) { return (jpto_builder *)calloc(1, sizeof(jpto_builder)); }

void jpto_builder_destroy(jpto_builder *b)
{
    if (!b) return;
    free(b->tris); free(b->nodes); free(b->inst); free(b->tlas); free(b->geom); free(b->data);
    free(b);
}

/* ---------------------------------------------------------------- std::nth_element (libstdc++) */

static inline int tri_less(const jpto_triangle *a, const jpto_triangle *b, int axis)
{
    return v4_get(&a->centroid, axis) < v4_get(&b->centroid, axis); /* C:175 */
}
static inline void tri_swap(jpto_triangle *a, jpto_triangle *b)
{
    jpto_triangle t = *a; *a = *b; *b = t;
}

/* bits/stl_heap.h: __push_heap / __adjust_heap / __make_heap / __pop_heap */
static void adjust_heap(jpto_triangle *first, ptrdiff_t hole, ptrdiff_t len, jpto_triangle value, int axis)
{
    const ptrdiff_t top = hole;
    ptrdiff_t second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (tri_less(&first[second], &first[second - 1], axis)) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    /* __push_heap */
    ptrdiff_t parent = (hole - 1) / 2;
    while (hole > top && tri_less(&first[parent], &value, axis)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void make_heap(jpto_triangle *first, ptrdiff_t len, int axis)
{
    if (len < 2) return;
    ptrdiff_t parent = (len - 2) / 2;
    for (;;) {
        jpto_triangle value = first[parent];
        adjust_heap(first, parent, len, value, axis);
        if (parent == 0) return;
        parent--;
    }
}
/* bits/stl_algo.h: __heap_select */
static void heap_select(jpto_triangle *first, jpto_triangle *middle, jpto_triangle *last, int axis)
{
    make_heap(first, middle - first, axis);
    for (jpto_triangle *i = middle; i < last; ++i)
        if (tri_less(i, first, axis)) {
            /* __pop_heap(first, middle, i) */
            jpto_triangle value = *i;
            *i = *first;
            adjust_heap(first, 0, middle - first, value, axis);
        }
}
/* __move_median_to_first */
static void move_median_to_first(jpto_triangle *result, jpto_triangle *a, jpto_triangle *b, jpto_triangle *c, int axis)
{
    if (tri_less(a, b, axis)) {
        if (tri_less(b, c, axis)) tri_swap(result, b);
        else if (tri_less(a, c, axis)) tri_swap(result, c);
        else tri_swap(result, a);
    } else if (tri_less(a, c, axis)) tri_swap(result, a);
    else if (tri_less(b, c, axis)) tri_swap(result, c);
    else tri_swap(result, b);
}
/* __unguarded_partition */
static jpto_triangle *unguarded_partition(jpto_triangle *first, jpto_triangle *last, jpto_triangle *pivot, int axis)
{
    for (;;) {
        while (tri_less(first, pivot, axis)) ++first;
        --last;
        while (tri_less(pivot, last, axis)) --last;
        if (!(first < last)) return first;
        tri_swap(first, last);
        ++first;
    }
}
/* __insertion_sort */
static void insertion_sort(jpto_triangle *first, jpto_triangle *last, int axis)
{
    if (first == last) return;
    for (jpto_triangle *i = first + 1; i != last; ++i) {
        if (tri_less(i, first, axis)) {
            jpto_triangle val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(jpto_triangle));
            *first = val;
        } else {
            /* __unguarded_linear_insert */
            jpto_triangle val = *i;
            jpto_triangle *lastp = i;
            jpto_triangle *next = i - 1;
            while (tri_less(&val, next, axis)) {
                *lastp = *next;
                lastp = next;
                --next;
            }
            *lastp = val;
        }
    }
}
static int lg2(ptrdiff_t n) { int k = 0; while (n > 1) { n >>= 1; k++; } return k; }

/* std::nth_element -> __introselect */
void jpto_nth_element_centroid(jpto_triangle *tris, int32_t first_i, int32_t nth_i, int32_t last_i, int32_t axis)
{
    jpto_triangle *first = tris + first_i, *nth = tris + nth_i, *last = tris + last_i;
    if (first == last || nth == last) return;
    int depth_limit = lg2(last - first) * 2;
    while (last - first > 3) {
        if (depth_limit == 0) {
            heap_select(first, nth + 1, last, axis);
            tri_swap(first, nth);
            return;
        }
        --depth_limit;
        /* __unguarded_partition_pivot */
        jpto_triangle *mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1, axis);
        jpto_triangle *cut = unguarded_partition(first + 1, last, first, axis);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(first, last, axis);
}

/* ---------------------------------------------------------------- BLAS build */

/* C:24-37 */
static bbox_t compute_bounding_box(const jpto_triangle *tris, int start, int end)
{
    bbox_t bbox = bbox_default();
    for (int i = start; i < end; i++)
        for (int j = 0; j < 3; j++) bbox_extend(&bbox, tris[i].vertices[j]);
    return bbox;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (hi < v ? hi : v); }

/* C:39-106 */
static float evaluate_sah(const jpto_triangle *tris, const jpto_bvh_node *node, int axis, float *bestSplit)
{
    enum { BINS = 8 };
    struct { bbox_t bounds; int count; } bins[BINS];
    for (int i = 0; i < BINS; i++) { bins[i].bounds = bbox_default(); bins[i].count = 0; }

    float minBound = v4_get(&node->aabbMin, axis);
    float maxBound = v4_get(&node->aabbMax, axis);
    float range = maxBound - minBound;
    if (range < 1e-6f) return 1e+30f;

    float invRange = 1.0f / range;

    for (uint32_t i = 0; i < node->tri_count; i++) {
        const jpto_triangle *tri = &tris[node->first_tri_index + i];
        float centroid = v4_get(&tri->centroid, axis);
        int binIdx = clampi((int)((float)BINS * (centroid - minBound) * invRange), 0, BINS - 1);
        bins[binIdx].count++;
        bbox_extend(&bins[binIdx].bounds, tri->vertices[0]);
        bbox_extend(&bins[binIdx].bounds, tri->vertices[1]);
        bbox_extend(&bins[binIdx].bounds, tri->vertices[2]);
    }

    float bestCost = 1e+30f;
    bbox_t leftAccum[BINS];
    int leftCount[BINS];

    bbox_t leftBox = bbox_default();
    int countLeft = 0;
    for (int i = 0; i < BINS - 1; i++) {
        bbox_extend(&leftBox, bins[i].bounds.min);
        bbox_extend(&leftBox, bins[i].bounds.max);
        countLeft += bins[i].count;
        leftAccum[i] = leftBox;
        leftCount[i] = countLeft;
    }

    bbox_t rightBox = bbox_default();
    int countRight = 0;
    for (int i = BINS - 1; i > 0; i--) {
        bbox_extend(&rightBox, bins[i].bounds.min);
        bbox_extend(&rightBox, bins[i].bounds.max);
        countRight += bins[i].count;
        float cost = bbox_area(&leftAccum[i - 1]) * (float)leftCount[i - 1] + bbox_area(&rightBox) * (float)countRight;

        if (cost < bestCost) {
            bestCost = cost;
            *bestSplit = minBound + ((float)i / (float)BINS) * range;
        }
    }
    return bestCost;
}

/* C:108-185 */
static uint32_t build_recursive(jpto_builder *b, int start, int end)
{
    if (start >= end) return 0;

    uint32_t node_index = (uint32_t)b->n_nodes;
    b->nodes = (jpto_bvh_node *)grow(b->nodes, &b->cap_nodes, b->n_nodes + 1, sizeof(jpto_bvh_node));
    b->n_nodes++;
    {
        jpto_bvh_node *node = &b->nodes[node_index];
        bbox_t bbox = compute_bounding_box(b->tris, start, end);
        node->aabbMin = bbox.min;
        node->aabbMax = bbox.max;
        node->left_child = 0;
        node->right_child = 0;
        node->first_tri_index = (uint32_t)start;
        node->tri_count = (uint32_t)(end - start);
    }
    jpto_bvh_node node = b->nodes[node_index];
    if (node.tri_count <= 4) return node_index;

    float bestSplit = 0.0f, bestCost = 1e30f;
    int bestAxis = -1;
    for (int axis = 0; axis < 3; axis++) {
        float split = 0.0f;
        float cost = evaluate_sah(b->tris, &node, axis, &split);
        if (cost < bestCost) {
            bestCost = cost;
            bestSplit = split;
            bestAxis = axis;
        }
    }

    vec4 e = v4_sub(node.aabbMax, node.aabbMin);
    float parentArea = e.x * e.y + e.y * e.z + e.z * e.x;
    float parentCost = (float)node.tri_count * parentArea;
    if (bestCost * 0.8f >= parentCost) return node_index;

    int i = start;
    int j = end - 1;
    while (i <= j) {
        float centroid = v4_get(&b->tris[i].centroid, bestAxis);
        if (centroid < bestSplit) i++;
        else { tri_swap(&b->tris[i], &b->tris[j]); j--; }
    }

    int left_count = i - start;
    if (left_count == 0 || left_count == (int)node.tri_count) {
        int mid = start + (end - start) / 2;
        jpto_nth_element_centroid(b->tris, start, mid, end, bestAxis);
        i = mid;
    }

    uint32_t l = build_recursive(b, start, i);
    b->nodes[node_index].left_child = l;
    uint32_t r = build_recursive(b, i, end);
    b->nodes[node_index].right_child = r;
    b->nodes[node_index].tri_count = 0;
    return node_index;
}

/* C:187-223 */
uint32_t jpto_builder_add_mesh(jpto_builder *b, const jpto_surface *surfaces, int32_t n_surfaces)
{
    int start = (int)b->n_tris;
    for (int32_t l = 0; l < n_surfaces; l++) {
        const jpto_surface *s = &surfaces[l];
        for (int32_t i = 0; i + 2 < s->n_indices; i += 3) {
            jpto_triangle tri;
            memset(&tri, 0, sizeof tri);
            for (int j = 0; j < 3; j++) {
                int32_t k = s->indices[i + j];
                tri.vertices[j] = v4(s->vertices[k * 3 + 0], s->vertices[k * 3 + 1], s->vertices[k * 3 + 2], 1.0f);
                tri.normals[j] = v4(s->normals[k * 3 + 0], s->normals[k * 3 + 1], s->normals[k * 3 + 2], 1.0f);
                tri.uvs[j][0] = s->uvs[k * 2 + 0];
                tri.uvs[j][1] = s->uvs[k * 2 + 1];
            }
            tri.materialIndex = (uint32_t)l;
            tri.centroid = v4_scale(v4_add(v4_add(tri.vertices[0], tri.vertices[1]), tri.vertices[2]), 0.33333333f);
            b->tris = (jpto_triangle *)grow(b->tris, &b->cap_tris, b->n_tris + 1, sizeof(jpto_triangle));
            b->tris[b->n_tris++] = tri;
        }
    }
    int end = (int)b->n_tris;
    return build_recursive(b, start, end);
}

/* ---------------------------------------------------------------- instances */

/* godot Basis::invert (cofactor form) + Transform3D::affine_inverse, float.  godot-cpp is an
 * un-vendored submodule (branch 4.3, not pinned); restated from its published algorithm.  This runs
 * before the path's boundary: it only produces the matrices in the BLASInstance record. */
void jpto_affine_inverse(const float *t, float *o)
{
    /* t[0..8] = basis rows, t[9..11] = origin */
#define R(r, c) t[(r) * 3 + (c)]
#define COFAC(r1, c1, r2, c2) (R(r1, c1) * R(r2, c2) - R(r1, c2) * R(r2, c1))
    float co0 = COFAC(1, 1, 2, 2), co1 = COFAC(1, 2, 2, 0), co2 = COFAC(1, 0, 2, 1);
    float det = R(0, 0) * co0 + R(0, 1) * co1 + R(0, 2) * co2;
    float s = 1.0f / det;
    float m[9];
    m[0] = co0 * s;               m[1] = COFAC(0, 2, 2, 1) * s; m[2] = COFAC(0, 1, 1, 2) * s;
    m[3] = co1 * s;               m[4] = COFAC(0, 0, 2, 2) * s; m[5] = COFAC(0, 2, 1, 0) * s;
    m[6] = co2 * s;               m[7] = COFAC(0, 1, 2, 0) * s; m[8] = COFAC(0, 0, 1, 1) * s;
#undef COFAC
#undef R
    for (int k = 0; k < 9; k++) o[k] = m[k];
    float nx = -t[9], ny = -t[10], nz = -t[11];
    o[9] = m[0] * nx + m[1] * ny + m[2] * nz;
    o[10] = m[3] * nx + m[4] * ny + m[5] * nz;
    o[11] = m[6] * nx + m[7] * ny + m[8] * nz;
}

/* U:15-37: Transform3D -> column-major float[16] */
static void transform_to_float(float *target, const float *t)
{
    /* basis.get_column(c) = (rows[0][c], rows[1][c], rows[2][c]) */
    for (int c = 0; c < 3; c++) {
        target[c * 4 + 0] = t[0 * 3 + c];
        target[c * 4 + 1] = t[1 * 3 + c];
        target[c * 4 + 2] = t[2 * 3 + c];
        target[c * 4 + 3] = 0.0f;
    }
    target[12] = t[9];
    target[13] = t[10];
    target[14] = t[11];
    target[15] = 1.0f;
}

/* H:73-115 */
uint32_t jpto_builder_add_instance(jpto_builder *b, uint32_t root, const float *transform12, const int32_t *material_ids,
                                   int32_t n_material_ids)
{
    jpto_blas_instance inst;
    memset(&inst, 0, sizeof inst); /* the reference leaves unset material[] uninitialised (H:73-79); zero here */
    inst.blas_index = root;
    int nm = n_material_ids < 3 ? n_material_ids : 3;
    for (int i = 0; i < nm; i++) inst.material[i] = (uint32_t)material_ids[i];

    float inv12[12];
    jpto_affine_inverse(transform12, inv12);
    transform_to_float(inst.transform, transform12);
    transform_to_float(inst.inverse_transform, inv12);

    /* update_aabb (H:90-115) */
    const jpto_bvh_node *node = &b->nodes[root];
    inst.aabbMin = v4(1e34f, 1e34f, 1e34f, 1.0f);
    inst.aabbMax = v4(-1e34f, -1e34f, -1e34f, 1.0f);
    vec4 bmin = node->aabbMin, bmax = node->aabbMax;
    for (int i = 0; i < 8; i++) {
        float corner[4] = {(i & 1) ? bmax.x : bmin.x, (i & 2) ? bmax.y : bmin.y, (i & 4) ? bmax.z : bmin.z, 1.0f};
        float tc[4] = {0.0f, 0.0f, 0.0f, 1.0f};
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) tc[j] += inst.transform[k * 4 + j] * corner[k];
        float s = 2.0f / tc[3];
        vec4 c = v4(tc[0] * s, tc[1] * s, tc[2] * s, tc[3] * s);
        inst.aabbMin = v4_min(inst.aabbMin, c);
        inst.aabbMax = v4_max(inst.aabbMax, c);
    }
    b->inst = (jpto_blas_instance *)grow(b->inst, &b->cap_inst, b->n_inst + 1, sizeof(jpto_blas_instance));
    b->inst[b->n_inst] = inst;
    return (uint32_t)b->n_inst++;
}

/* ---------------------------------------------------------------- TLAS */

/* C:319-340 */
static int find_best_match(const jpto_tlas_node *nodes, const int *list, int N, int A)
{
    float smallest = 1e30f;
    int bestB = -1;
    for (int B = 0; B < N; B++) {
        if (B != A) {
            const jpto_tlas_node *a = &nodes[list[A]], *bb = &nodes[list[B]];
            float bmaxx = smax(a->aabbMax[0], bb->aabbMax[0]), bmaxy = smax(a->aabbMax[1], bb->aabbMax[1]),
                  bmaxz = smax(a->aabbMax[2], bb->aabbMax[2]);
            float bminx = smin(a->aabbMin[0], bb->aabbMin[0]), bminy = smin(a->aabbMin[1], bb->aabbMin[1]),
                  bminz = smin(a->aabbMin[2], bb->aabbMin[2]);
            float ex = bmaxx - bminx, ey = bmaxy - bminy, ez = bmaxz - bminz;
            float surfaceArea = ex * ey + ey * ez + ez * ex;
            if (surfaceArea < smallest) {
                smallest = surfaceArea;
                bestB = B;
            }
        }
    }
    return bestB;
}

/* C:264-317 */
static void tlas_build(jpto_builder *b)
{
    int blasCount = (int)b->n_inst;
    free(b->tlas);
    size_t cap = (size_t)(blasCount > 0 ? blasCount * 2 : 1);
    b->tlas = (jpto_tlas_node *)calloc(cap, sizeof(jpto_tlas_node));
    b->n_tlas = 0;
    b->n_tlas++; /* slot 0 reserved for the root */
    if (blasCount == 0) return;

    if (blasCount <= 0 || blasCount > 65535) return;
    int *nodeIdx = (int *)malloc(sizeof(int) * (size_t)(unsigned)blasCount);
    int nodesUsed = 1;
    for (int i = 0; i < blasCount; i++) {
        jpto_tlas_node node;
        memset(&node, 0, sizeof node);
        node.aabbMin[0] = b->inst[i].aabbMin.x; node.aabbMin[1] = b->inst[i].aabbMin.y; node.aabbMin[2] = b->inst[i].aabbMin.z;
        node.aabbMax[0] = b->inst[i].aabbMax.x; node.aabbMax[1] = b->inst[i].aabbMax.y; node.aabbMax[2] = b->inst[i].aabbMax.z;
        node.blas = (uint32_t)i;
        node.leftRight = 0;
        b->tlas[b->n_tlas++] = node;
        nodeIdx[i] = nodesUsed++;
    }

    int A = 0, B = find_best_match(b->tlas, nodeIdx, blasCount, A);
    while (blasCount > 1) {
        int C = find_best_match(b->tlas, nodeIdx, blasCount, B);
        if (A == C) {
            int nodeIdxA = nodeIdx[A], nodeIdxB = nodeIdx[B];
            const jpto_tlas_node *nodeA = &b->tlas[nodeIdxA];
            const jpto_tlas_node *nodeB = &b->tlas[nodeIdxB];
            jpto_tlas_node newNode;
            memset(&newNode, 0, sizeof newNode); /* .blas of internal nodes is uninitialised in the reference */
            newNode.leftRight = (uint32_t)nodeIdxA + ((uint32_t)nodeIdxB << 16);
            for (int k = 0; k < 3; k++) {
                newNode.aabbMin[k] = smin(nodeA->aabbMin[k], nodeB->aabbMin[k]);
                newNode.aabbMax[k] = smax(nodeA->aabbMax[k], nodeB->aabbMax[k]);
            }
            b->tlas[b->n_tlas++] = newNode;
            nodeIdx[A] = nodesUsed++;
            nodeIdx[B] = nodeIdx[--blasCount];
            B = find_best_match(b->tlas, nodeIdx, blasCount, A);
        } else {
            A = B;
            B = C;
        }
    }
    b->tlas[0] = b->tlas[nodeIdx[A]];
    free(nodeIdx);
}

/* G:350-365 */
void jpto_builder_finish(jpto_builder *b)
{
    tlas_build(b);
    free(b->geom);
    free(b->data);
    b->geom = (jpto_tri_geometry *)calloc(b->n_tris ? b->n_tris : 1, sizeof(jpto_tri_geometry));
    b->data = (jpto_tri_data *)calloc(b->n_tris ? b->n_tris : 1, sizeof(jpto_tri_data));
    for (size_t i = 0; i < b->n_tris; i++) {
        const jpto_triangle *t = &b->tris[i];
        for (int k = 0; k < 3; k++) b->geom[i].vertices[k] = t->vertices[k];
        jpto_tri_data *d = &b->data[i];
        d->n0[0] = t->normals[0].x; d->n0[1] = t->normals[0].y; d->n0[2] = t->normals[0].z;
        d->material_index = t->materialIndex;
        d->n1 = t->normals[1];
        d->n2 = t->normals[2];
        for (int k = 0; k < 3; k++) { d->uvs[k][0] = t->uvs[k][0]; d->uvs[k][1] = t->uvs[k][1]; }
    }
}

uint32_t jpto_builder_counts(const jpto_builder *b, uint32_t *n_tri, uint32_t *n_nodes, uint32_t *n_inst, uint32_t *n_tlas)
{
    if (n_tri) *n_tri = (uint32_t)b->n_tris;
    if (n_nodes) *n_nodes = (uint32_t)b->n_nodes;
    if (n_inst) *n_inst = (uint32_t)b->n_inst;
    if (n_tlas) *n_tlas = (uint32_t)b->n_tlas;
    return (uint32_t)b->n_tris;
}
const jpto_triangle *jpto_builder_triangles(const jpto_builder *b) { return b->tris; }
const jpto_tri_geometry *jpto_builder_tri_geom(const jpto_builder *b) { return b->geom; }
const jpto_tri_data *jpto_builder_tri_data(const jpto_builder *b) { return b->data; }
const jpto_bvh_node *jpto_builder_nodes(const jpto_builder *b) { return b->nodes; }
const jpto_blas_instance *jpto_builder_instances(const jpto_builder *b) { return b->inst; }
const jpto_tlas_node *jpto_builder_tlas(const jpto_builder *b) { return b->tlas; }
