"""Known-answer vectors for FPS and ball query, derived BY HAND from the text of the reference kernels
(vgtk/vgtk/cuda/grouping_cuda_kernel.cu:67-113 ball query, :339-466 FPS incl. the `__update` reduction tree and the
block size rule :29-33) -- independently of oracle/epn_oracle.c, which they pin, and of the HIP kernels.  All
coordinates are small integers (or far from every threshold), so every distance is exact in fp32 and no expected
value depends on the floating-point evaluation order.  The derivations are spelled out next to each vector.
"""
import numpy as np


def line(xs):
    """Points on the x axis -> [1, 3, n] float32."""
    p = np.zeros((1, 3, len(xs)), dtype=np.float32)
    p[0, 0] = xs
    return p


FPS = []

# A. n = 8 -> block_size = 2^floor(log2 8) = 8, one point per thread.  x = [1,2,4,8,16,3,5,10], m = 6.
#  r1 old=0 (x=1): temp = (x-1)^2 = [0,1,9,49,225,4,16,81]                      -> max 225 @4
#  r2 old=4 (x=16): d = [225,196,144,64,0,169,121,36], temp = [0,1,9,49,0,4,16,36] -> max 49 @3
#  r3 old=3 (x=8): d = [49,36,16,0,64,25,9,4], temp = [0,1,9,0,0,4,9,4]: 9 at k=2 AND k=6.
#     tree (tid, tid+4): t0 (0,i0)/(0,i4)->i0; t1 (1,i1)/(4,i5)->i5; t2 (9,i2)/(9,i6) tie keeps idx1 -> i2; t3 (0,i3)/(4,i7)->i7
#     (tid, tid+2): t0 (0,i0)/(9,i2)->i2; t1 (4,i5)/(4,i7) tie -> i5;  (0,1): 9 vs 4 -> i2                -> 2
#  r4 old=2 (x=4): d = [9,4,0,16,144,1,1,36], temp = [0,1,0,0,0,1,1,4]           -> max 4 @7
#  r5 old=7 (x=10): d = [81,64,36,4,36,49,25,0], temp = [0,1,0,0,0,1,1,0]: 1 at k = 1, 5, 6.
#     (tid, tid+4): t0 (0,i0)/(0,i4)->i0; t1 (1,i1)/(1,i5) tie -> i1; t2 (0,i2)/(1,i6)->i6; t3 (0,i3)/(0,i7)->i3
#     (tid, tid+2): t0 (0,i0)/(1,i6)->i6; t1 (1,i1)/(0,i3)->i1;  (0,1): (1,i6)/(1,i1) tie keeps idx1 -> i6  -> 6
#     (the tie goes to index 6, NOT to the lowest index 1: this pins the tree order)
FPS.append(dict(name="ties_follow_the_reduction_tree", xyz=line([1, 2, 4, 8, 16, 3, 5, 10]), m=6,
                idx=[0, 4, 3, 2, 7, 6]))

# B. dead zone |p|^2 <= 1e-3 (:385-387): such points never update temp and contribute (best, besti) = (-1, 0).
#  k0 (1,0,0)  k1 (.01,.01,.01) |p|^2=3e-4 dead  k2 (0,2,0)  k3 (0,0,0) dead  k4 (0,0,3)  k5 (-1,0,0)
#  k6 (.03,0,0) |p|^2=9e-4 dead  k7 (0,-2,0);  n = 8 -> block 8.
#  r1 old=0: d = k0 0, k2 5, k4 10, k5 4, k7 5; dists = [0,-1,5,-1,10,4,-1,5]                              -> 4
#  r2 old=4 (0,0,3): temp = k0 0, k2 min(5,13)=5, k4 0, k5 min(4,10)=4, k7 min(5,13)=5
#     (tid,tid+4): t0 (0,i0)/(0,i4)->i0; t1 (-1,i0)/(4,i5)->i5; t2 (5,i2)/(-1,i0)->i2; t3 (-1,i0)/(5,i7)->i7
#     (tid,tid+2): t0 (0,i0)/(5,i2)->i2; t1 (4,i5)/(5,i7)->i7;  (0,1): (5,i2)/(5,i7) tie -> i2                -> 2
#  r3 old=2 (0,2,0): temp = k0 0, k2 0, k4 0, k5 min(4,5)=4, k7 min(5,16)=5                                 -> 7
#  r4 old=7: temp = k5 min(4,5)=4, rest 0                                                                   -> 5
#  r5 old=5: every live temp is 0, dead threads carry -1: (tid,tid+4): t0 i0, t1 (-1,i0)/(0,i5)->i5, t2 i2,
#     t3 (-1,i0)/(0,i7)->i7; (tid,tid+2): t0 (0,i0)/(0,i2)->i0, t1 (0,i5)/(0,i7)->i5; (0,1): tie -> i0       -> 0
_b = np.array([[1, 0, 0], [.01, .01, .01], [0, 2, 0], [0, 0, 0], [0, 0, 3], [-1, 0, 0], [.03, 0, 0], [0, -2, 0]],
              dtype=np.float32)
FPS.append(dict(name="dead_zone_points_are_skipped", xyz=np.ascontiguousarray(_b.T[None]), m=6, idx=[0, 4, 2, 7, 5, 0]))

# C. n = 6 is not a power of two: block_size = 2^floor(log2 6) = 4 (:29-33); thread t owns k = t, t+4: thread 0 -> {0,4},
#  thread 1 -> {1,5}, thread 2 -> {2}, thread 3 -> {3}; inside a thread `besti = d2 > best ? k : besti` keeps the FIRST
#  maximum.  x = [1,3,6,10,2,7], m = 4.
#  r1 old=0 (x=1): d = [0,4,25,81,1,36]: t0 (4,1) t1 (5,36) t2 (2,25) t3 (3,81); (tid,tid+2): t0 -> (2,25), t1 -> (3,81) -> 3
#  r2 old=3 (x=10): temp = [0,4,16,0,1,9]: t0 (4,1) t1 (5,9) t2 (2,16) t3 (3,0); t0 -> (2,16), t1 -> (5,9)        -> 2
#  r3 old=2 (x=6): d = [25,9,0,16,16,1], temp = [0,4,0,0,1,1]: t0 (4,1) t1 (1,4) [k5: 1 > 4 no] t2 (2,0) t3 (3,0);
#     t0 (1,i4)/(0,i2)->i4, t1 (4,i1)/(0,i3)->i1; (0,1): 1 vs 4 -> i1                                         -> 1
FPS.append(dict(name="block_size_is_a_power_of_two_below_n", xyz=line([1, 3, 6, 10, 2, 7]), m=4, idx=[0, 3, 2, 1]))


BALLQ = []
_support = line([0, 1, 2, 3, 4, 5, 6, 7])


def _q(xs):
    return line(xs)


# radius 1.6 (r^2 = 2.56, strict <), nsample 4, zero-initialised idx (grouping_cuda.cpp:80-82):
#  q x=0   : hits k=0 (0), k=1 (1); k=2 (4) out -> cnt 2 < nsample-1 = 3 -> cyclic fill idx[2]=idx[0], idx[3]=idx[1] -> [0,1,0,1]
#  q x=3.5 : hits 2 (2.25), 3 (.25), 4 (.25), 5 (2.25) in index order, cnt = 4 = nsample                    -> [2,3,4,5]
#  q x=100 : no hit, cnt 0 < 3 -> idx[k] = idx[k] (self copies of the zero fill)                              -> [0,0,0,0]
BALLQ.append(dict(name="fill_and_full", query=_q([0, 3.5, 100]), support=_support, radius=1.6, nsample=4,
                  idx=[[0, 1, 0, 1], [2, 3, 4, 5], [0, 0, 0, 0]]))
# radius 1.1 (r^2 = 1.21): q x=1 hits 0 (1), 1 (0), 2 (1) -> cnt 3 == nsample-1 -> NO fill, last slot keeps the zero -> [0,1,2,0]
BALLQ.append(dict(name="cnt_equals_nsample_minus_1_is_not_filled", query=_q([1]), support=_support, radius=1.1, nsample=4,
                  idx=[[0, 1, 2, 0]]))
# radius 2.0 (r^2 = 4 exactly): q x=0: k=2 has d2 = 4, NOT < 4 -> hits 0, 1 only -> cnt 2 < 3 -> [0,1,0,1]
BALLQ.append(dict(name="strict_inequality", query=_q([0]), support=_support, radius=2.0, nsample=4, idx=[[0, 1, 0, 1]]))
# radius 10: q x=3.5: everything is inside; the scan stops at cnt == nsample -> the FIRST four indices        -> [0,1,2,3]
BALLQ.append(dict(name="first_nsample_hits_in_index_order", query=_q([3.5]), support=_support, radius=10.0, nsample=4,
                  idx=[[0, 1, 2, 3]]))


# ---- fp64 coordinates only.  The templated reference kernel keeps `float radius` as its parameter and squares it in FLOAT
#  (`scalar_t radius2 = radius * radius;`, grouping_cuda_kernel.cu:67,80): with scalar_t = double the threshold is the float
#  product widened, NOT the double product of a double radius.  radius = 0.2:
#    float(0.2)                 = 0.20000000298023224
#    its exact square           = 0.04000000119209290...; the float32 neighbours are 0.03999999910593033 and
#                                 0.04000000283122063 (spacing 3.7e-9): the product rounds to 0.04000000283122063
#    threshold as the reference has it   r2 = 0.04000000283122063
#    threshold of a double radius        0.2 * 0.2 = 0.04000000000000001   (what round 3's f64 path used)
#  support k=1 at x = 0.2000000025: d2 = 0.04000000100000000625 -- BETWEEN the two.  Reference: d2 < r2 -> hit.
#  q x=0, support x = [0, 0.2000000025, 5], nsample 4: hits k=0 (d2 = 0), k=1 -> cnt 2 < 3 -> cyclic fill -> [0,1,0,1]
#  (a double-squared radius would give one hit: [0,0,0,0]).
BALLQ_F64 = []
_s64 = np.zeros((1, 3, 3), dtype=np.float64)
_s64[0, 0] = [0.0, 0.2000000025, 5.0]
BALLQ_F64.append(dict(name="radius_is_squared_in_float_then_widened", query=np.zeros((1, 3, 1), dtype=np.float64), support=_s64,
                      radius=0.2, nsample=4, idx=[[0, 1, 0, 1]]))
# the other side: x = 0.20000000715 -> d2 = 0.0400000028600000511 > r2 = 0.04000000283122063 -> only k=0 -> cnt 1 -> [0,0,0,0]
_s64b = _s64.copy()
_s64b[0, 0, 1] = 0.20000000715
BALLQ_F64.append(dict(name="just_outside_the_float_squared_radius", query=np.zeros((1, 3, 1), dtype=np.float64), support=_s64b,
                      radius=0.2, nsample=4, idx=[[0, 0, 0, 0]]))
