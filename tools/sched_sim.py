"""Monte-Carlo model of how one wave of the persistent kernel spends its time, used to rank
scheduling policies before building them (development aid, not part of the library).

Model: 64 lanes; a ray needs L traversal steps, L = 0 with probability 1/2 (the ray misses
the bounds of the mesh and only meets the hoisted ground sphere) and 4 + Geometric(mean 11.4)
otherwise (mean 7.7, the dragon's measured node + leaf steps per ray); a traversal step costs
1 whatever the number of lanes in it; a shade / regenerate round costs ROUND = 8.07 steps
whatever the number of lanes in it (6876 against 852 wave cycles, instrumented kernel).
Policies:
  single   the kernel as built: a round when `thr` lanes wait
  double   two paths per lane, the second parked until the first is done
  pool     any idle lane takes any ready ray of a pool of P paths per wave; all finished
           paths are shaded in rounds of 64
Output: rays per step-unit (higher is better).
"""
import numpy as np
rng = np.random.default_rng(1)
def newlen(n):
    z = rng.random(n) < 0.5
    L = 4 + rng.geometric(1/11.4, n)
    return np.where(z, 0, L)
ROUND = 8.07
def sim_single(thr, iters=20000):
    rem = newlen(64); steps = 0.0; rays = 64
    t = ROUND
    for _ in range(iters):
        trav = rem > 0
        if (~trav).sum() >= thr or not trav.any():
            k = (~trav).sum(); rem[~trav] = newlen(k); rays += k; t += ROUND; continue
        rem[trav] -= 1; t += 1
    return rays / t
def sim_double(thr, swapcost=0.5, iters=20000, second_min=1):
    # resident rem; parked: length (ready, >=0) or -1 = done/none
    rem = newlen(64); pk = newlen(64); rays = 128; t = 2 * ROUND
    pk_ready = np.ones(64, bool)
    for _ in range(iters):
        trav = rem > 0
        can_swap = (~trav) & pk_ready
        if can_swap.sum() >= 8 or (can_swap.any() and not trav.any()):
            rem[can_swap] = pk[can_swap]; pk_ready[can_swap] = False; t += swapcost; continue
        idle = (~trav) & (~pk_ready)
        if idle.sum() >= thr or not (trav.any() or can_swap.any()):
            # round half 1: residents of idle lanes; half 2: their parked
            k = idle.sum(); rem[idle] = newlen(k); pk[idle] = newlen(k); pk_ready[idle] = True
            rays += 2 * k; t += 2 * ROUND + swapcost; continue
        rem[trav] -= 1; t += 1
    return rays / t
def sim_pool(P, thr_refill=8, thr_round=40, iters=20000):
    # pool of P paths per wave; lanes take any ready ray; round shades all done in batches of 64
    rem = np.zeros(64, int); ready = list(newlen(P)); done = 0; rays = P; t = (P / 64) * ROUND
    # zero-length rays are done immediately at no traversal cost
    def take():
        nonlocal done
        while ready:
            L = ready.pop()
            if L == 0: done += 1; continue
            return L
        return 0
    for _ in range(iters):
        trav = rem > 0
        idle = (~trav).sum()
        if ready and idle >= thr_refill:
            for i in np.where(~trav)[0]:
                L = take()
                if L == 0: break
                rem[i] = L
            t += 0.5; continue
        if (not ready) and (idle >= thr_round or not trav.any()):
            # everything not travelling is done: shade in batches of 64
            n_done = P - trav.sum()
            t += np.ceil(n_done / 64) * ROUND
            ready = list(newlen(n_done)); rays += n_done; continue
        rem[trav] -= 1; t += 1
    return rays / t
for thr in (24, 32, 40, 48): print("single thr", thr, "rays per step-unit %.3f" % sim_single(thr))
for thr in (24, 32, 40, 48): print("double thr", thr, "%.3f" % sim_double(thr))
for P in (128, 256, 512):
    for tr in (24, 40): print("pool", P, tr, "%.3f" % sim_pool(P, 8, tr))
print("ideal", 64 / (7.7 + ROUND))
def sim_single2(thr, thr2, iters=20000):
    rem = newlen(64); rays = 64; t = ROUND; after_round = True
    for _ in range(iters):
        trav = rem > 0
        idle = (~trav).sum()
        if idle >= thr or not trav.any() or (after_round and idle >= thr2):
            rem[~trav] = newlen(idle); rays += idle; t += ROUND; after_round = True; continue
        after_round = False
        rem[trav] -= 1; t += 1
    return rays / t
print("--- immediate re-round")
for thr2 in (12, 16, 20, 24, 28, 32): print("thr 40, re-round at", thr2, "%.3f" % sim_single2(40, thr2))
# sensitivity: cheaper rounds
for R in (8.07, 6.0, 4.0):
    ROUND = R
    print("ROUND", R, "best single:", max((sim_single(t), t) for t in (16, 24, 32, 40, 48)))
