"""Closer model of the several-paths-per-lane design (see lane_sim.py for the idea and the numbers' sources).
Differences: a lane loads at most ONE path per trip (a swap through LDS, charged to the wave when any lane swaps) and keeps
working on its current path while that path can act; finished paths retire IN SAMPLE ORDER (a finished path blocks its slot
until every earlier sample of the pixel has retired: Renderer.cpp:66 sums in order); blocks run when K lanes can use them or a
lane would otherwise idle. usage: lane_sim2.py"""
import random, sys
COST = dict(trace=148, V=170, N=92, Lt=100, B=110, T=68)
P_DIE, P_T, P_LOBE = 0.6475, 0.126, 0.2265
Q_SHADOW, P_HIT = 0.43, 0.97

class Path:
    __slots__ = ('state', 'seq')
    def __init__(s, seq): s.state, s.seq = 'N', seq   # N: wants camera ray; ray/shadow: has a ray; Lt/B/T: wants block; done: finished

def simulate(P, K, swap_cost, paths_per_lane=400, seed=1, n_always=False, S=8):
    rnd = random.Random(seed)
    L = 64
    nxt = [0] * L            # next sample sequence number to start
    retire = [0] * L         # next sequence number to retire
    slots = [[None] * P for _ in range(L)]
    cur = [0] * L
    cost = trips = busy = swaps_trips = 0
    finished = 0
    def refill(i):
        for j in range(P):
            if slots[i][j] is None and nxt[i] < paths_per_lane:
                slots[i][j] = Path(nxt[i]); nxt[i] += 1
    def do_retire(i):
        nonlocal finished
        progress = True
        while progress:
            progress = False
            for j in range(P):
                p = slots[i][j]
                if p is not None and p.state == 'done' and p.seq == retire[i]:
                    slots[i][j] = None; retire[i] += 1; finished += 1; progress = True
        refill(i)
    for i in range(L): refill(i)
    while True:
        alive = [i for i in range(L) if any(s is not None for s in slots[i])]
        if not alive: break
        trips += 1
        def wants(i, b): return any(p is not None and p.state == b for p in slots[i])
        def runnable(i): return any(p is not None and p.state in ('ray', 'shadow') for p in slots[i])
        run = set()
        for b in ('N', 'T', 'Lt', 'B'):
            d = sum(1 for i in alive if wants(i, b))
            if d and (d >= K or (b == 'N' and n_always)):
                run.add(b)
        # lanes that could do nothing this trip: while there are S of them (or all that are left), run the block most of them want
        while True:
            idle = [i for i in alive if not runnable(i) and not any(wants(i, b) for b in run)]
            if len(idle) < min(S, len(alive)) or not idle:
                break
            best = max(('N', 'T', 'Lt', 'B'), key=lambda b: sum(1 for i in idle if wants(i, b)))
            if best in run or not any(wants(i, best) for i in idle):
                break
            run.add(best)
        if trips > 400000:
            raise RuntimeError('no progress')
        swapped = False
        ran = 0
        for i in alive:
            # choose the path to work on: keep the current one if it can act, else prefer a path whose block runs now, else a ray
            def can_act(p): return p is not None and (p.state in ('ray', 'shadow') or p.state in run)
            j = cur[i]
            if not can_act(slots[i][j]):
                cands = [k for k in range(P) if can_act(slots[i][k])]
                if not cands: continue
                pref = [k for k in cands if slots[i][k].state in run]
                j = (pref or cands)[0]
                if P > 1: swapped = True
                cur[i] = j
            p = slots[i][j]
            # chain of actions within the trip
            if p.state == 'N': p.state = 'ray'
            elif p.state == 'T': p.state = 'ray'
            elif p.state == 'Lt':
                if rnd.random() < Q_SHADOW: p.state = 'shadow'
                else: p.state = 'B' if 'B' not in run else 'ray'
            elif p.state == 'B': p.state = 'ray'
            if p.state == 'B': continue
            ran += 1
            if p.state == 'shadow':
                p.state = 'B'
                if 'B' in run: p.state = 'ray'  # (B runs after the vertex block in the same trip, as now)
                continue
            # extension / camera ray: vertex
            if rnd.random() > P_HIT: p.state = 'done'
            else:
                u = rnd.random()
                p.state = 'done' if u < P_DIE else ('T' if u < P_DIE + P_T else 'Lt')
            if p.state == 'done':
                do_retire(i)
        cost += COST['trace'] + COST['V'] + sum(COST[b] for b in run) + (swap_cost if swapped else 0)
        busy += ran
    return cost / finished * 64, trips, busy / (trips * 64)

if __name__ == '__main__':
    base, _, e = simulate(1, 1, 0)
    print('1 path per lane: %.0f wave-instructions per 64 paths, lane slots used %.3f (measured: 1243 and 0.958 with pass stealing)' % (base, e))
    for P in (2, 3, 4):
        for K in (24, 32, 40):
            for S in (4, 8, 16):
                for na in (False, True):
                    c, t, e = simulate(P, K, 45, n_always=na, S=S)
                    print('paths/lane %d K=%2d idle-lane trigger %2d swap 45 %s: %.3f of baseline, lane slots used %.3f' % (P, K, S, 'N every trip ' if na else 'N deferred   ', c / base, e))
