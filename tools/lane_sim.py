"""Monte-Carlo model of one wave of the megakernel: what would SEVERAL PATHS PER LANE with parked states buy?
A lane owns P path slots. Every trip it advances ONE runnable path (ray ready): trace + vertex block; a path that then wants
a low-occupancy block (new camera ray N, light sample Lt, BSDF sample B, transmission T) is PARKED until the wave executes that
block; the wave executes a block when at least K lanes have a path parked for it (or when a lane has nothing else to run).
Costs are the FAST kernel's wave-instructions per block; branching probabilities are spheres.json's
(device counters: 1.532 vertices, 0.347 light samples, 0.193 transmissions, 0.149 shadow rays per camera path).
usage: lane_sim.py [paths_per_lane ...]"""
import random, sys
COST = dict(trace=148, V=170, N=92, Lt=100, B=110, T=68, swap=0)
P_DIE, P_T, P_LOBE = 0.6475, 0.126, 0.2265
Q_SHADOW, P_HIT = 0.43, 0.97

def simulate(P, K, swap_cost, n_paths_per_lane=400, seed=1):
    rnd = random.Random(seed)
    L = 64
    # path state: 'N' wants camera ray; 'ray' has a ray (kind ext/shadow); 'Lt','B','T' parked; None = slot exhausted
    slots = [['N'] * P for _ in range(L)]
    kind = [['ext'] * P for _ in range(L)]
    remaining = [n_paths_per_lane] * L  # camera paths still to start
    done_paths = 0
    cost = 0
    trips = 0
    busy_lanes = 0
    last = [0] * L
    while True:
        # which lanes still have work
        alive = [i for i in range(L) if any(s is not None for s in slots[i])]
        if not alive:
            break
        trips += 1
        # --- block execution decisions: count parked demand
        def demand(b):
            return sum(1 for i in alive if b in slots[i])
        starving = [i for i in alive if not any(s == 'ray' for s in slots[i])]
        for b in ('N', 'T', 'Lt', 'B'):
            d = demand(b)
            if d == 0:
                continue
            need = any(b in slots[i] for i in starving)
            if d >= K or need:
                cost += COST[b]
                for i in alive:
                    for j in range(P):
                        if slots[i][j] == b:
                            if b == 'N':
                                if remaining[i] > 0:
                                    remaining[i] -= 1
                                    slots[i][j], kind[i][j] = 'ray', 'ext'
                                else:
                                    slots[i][j] = None
                            elif b == 'T':
                                slots[i][j], kind[i][j] = 'ray', 'ext'
                            elif b == 'Lt':
                                if rnd.random() < Q_SHADOW:
                                    slots[i][j], kind[i][j] = 'ray', 'shadow'
                                else:
                                    slots[i][j] = 'B'  # (executed below in the same trip if B runs)
                            elif b == 'B':
                                slots[i][j], kind[i][j] = 'ray', 'ext'
                            if b != 'Lt':
                                break  # one path per lane per block execution
        # --- trace + vertex: every lane advances one runnable path
        ran = 0
        for i in alive:
            c = [j for j in range(P) if slots[i][j] == 'ray']
            if not c:
                continue
            j = c[(last[i] + 1) % len(c)] if len(c) > 1 else c[0]
            if P > 1 and j != last[i]:
                cost_swap[0] += 1
            last[i] = j
            ran += 1
            if kind[i][j] == 'shadow':
                slots[i][j] = 'B'
                continue
            if rnd.random() > P_HIT:
                slots[i][j] = 'N'; done_paths += 1; continue
            u = rnd.random()
            if u < P_DIE:
                slots[i][j] = 'N'; done_paths += 1
            elif u < P_DIE + P_T:
                slots[i][j] = 'T'
            else:
                slots[i][j] = 'Lt'
        cost += COST['trace'] + COST['V'] + (swap_cost if P > 1 else 0)
        busy_lanes += ran
    return cost / done_paths * 64, trips, busy_lanes / (trips * 64)

cost_swap = [0]
if __name__ == '__main__':
    base = None
    for P in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
        for K in ((1,) if P == 1 else (1, 16, 24, 32, 40, 48)):
            for swap in ((0,) if P == 1 else (0, 35, 50)):
                c, trips, eff = simulate(P, K, swap)
                if base is None:
                    base = c
                print('paths/lane %d  threshold K=%2d  swap cost %2d: %.0f wave-instructions per 64 paths (%.3f of baseline), lane slots used %.3f' % (P, K, swap, c, c / base, eff))
