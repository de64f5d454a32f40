// deferred.inc.hip -- EXPERIMENT (round 3, SURVEY row N1 "path compaction"; measured slower, not part of the product library):
// the render loop with DEFERRED light/BSDF sampling. Included by integrator.inc.hip when built -DKAJO_WITH_DEFERRED
// (`make -C kajo_amd/csrc experiments` -> kajo_amd/libkajo_hip_exp.so; selected per handle with KAJO_FLAG_DEFERRED). It uses the
// product's own device functions (trace, BSDFs, light sampling, generator); only the loop around them differs.
// STRICT = oracle bit for bit (tests/test_experiments_gpu.py). What it measured: DESIGN.md section 8.
namespace
{

enum : int
{
    DM_FREE = 0,        // the lane's registers hold no path: it may start a camera path or resume a parked vertex
    DM_EXTEND = 1,      // the ray in (O, d) continues the path: shade what it hits
    DM_SHADOW_MID = 2,  // the ray asks whether a light is visible; more lights follow (the vertex stays parked)
    DM_SHADOW_LAST = 3, // ... the last light of the vertex: the BSDF-sampled extension ray is already waiting in d2
    DM_RETIRE = 4,      // the path is complete, L is its radiance
    DM_DONE = 5
};

// ---- the render loop ---------------------------------------------------------------------------------------------------
// One lane = one pixel; the lane works through that pixel's n*n*passes camera paths in the reference's order, so the
// per-pixel float sums are formed exactly as Renderer.cpp:66-71 forms them. Every trip round the loop traces ONE ray per
// lane through the whole scene. What a lane does before and after that ray is one of two kinds of work:
//
//   E work   a camera ray for a new path (Renderer.cpp:51-64), and the vertex the ray finds: emission, Russian roulette,
//            the lobe coins, refraction (Shader.cpp:113-178). Two thirds of all paths end right there.
//   L work   for a vertex that survived: light sampling with its shadow ray and BSDF sampling (Shader.cpp:50-86,180-215).
//
// Round 2's kernel ran both kinds in every trip, each under its own exec mask: the L blocks -- a third of the
// instructions -- executed with a quarter of the lanes. Here a surviving vertex is PARKED: its state (20-odd dwords) goes to
// a per-lane FIFO in LDS (the "stash"), the lane's registers are free again and the lane starts its pixel's next camera
// path in the very next trip. The wave runs the L blocks only in trips where enough lanes have a parked vertex
// (RenderArgs::thrL) or are stuck without one being resumed (thrStall); every such lane then takes its oldest vertex back
// and runs the blocks together with the others. The light sample and the BSDF sample of a vertex are drawn in ONE visit
// (their random numbers do not depend on what the shadow ray finds: the stream order light, then BSDF, is kept): the
// lane traces the shadow ray in that trip and the extension ray in the next, so a parked vertex needs exactly one visit of
// the L blocks per light that has to be traced.
//
// Paths of one pixel therefore complete out of order. Their radiances enter the pixel's sum IN SAMPLE ORDER all the
// same: a path carries its sequence number; one that completes while an older path of the lane is still parked leaves its
// radiance in a small per-lane ring in LDS, and the lane adds the ring entries when the older path retires. A lane never
// has more than ringSlots + 1 paths in flight, nor samples of more than two passes.
//
// KAT (known-answer mode): instead of its pixel's camera paths a lane runs ONE path from a given ray
// and RNG state and reports its radiance and the RNG state it ends in (kajo_hip_kat_shade).
//
// SPLIT (small frames): the waves of a workgroup share ONE 8x8-pixel block and divide the launch's passes among
// them, so that a frame with fewer blocks than the chip has wave slots still fills it. Every pass's term radiance / S
// goes to an LDS table [pass][pixel]; after a barrier wave 0 adds the terms to the accumulation in pass order -- the
// float sums are those of one wave doing all the passes.
#define KAJO_STASH_QUADS 6
template <bool COLD_LDS, bool KAT, bool SPLIT = false>
KDEV void renderBodyDeferred(const RenderArgs& args, unsigned char* ldsRaw)
{
    const DSceneView& sc = args.scene;
    const int np = sc.nPlanes;

    // ---- stage the scene into LDS (one copy per workgroup) ------------------------------------
    const LdsScene lds = stageToLds<COLD_LDS>(sc, ldsRaw);

    // ---- per-wave LDS: [mailbox: 64 x stealWindow float4][ring: ringSlots x 3 x 64 float][stash: depth x 6 x 64 float4]
    const int lane = threadIdx.x & 63;
    const int stealWindow = args.stealWindow;
    unsigned char* waveLds = ldsRaw + args.perWaveOffset + (threadIdx.x >> 6) * args.perWaveBytes;
    DFloat4* mailbox = reinterpret_cast<DFloat4*>(waveLds); // [lane][stealWindow]: passes rendered for this lane's pixel by others
    float* ring = reinterpret_cast<float*>(waveLds + args.ringOffset) + lane;        // [slot][3][64]
    DFloat4* stash = reinterpret_cast<DFloat4*>(waveLds + args.stashOffset) + lane;  // [entry][KAJO_STASH_QUADS][64]
    const int stashDepth = args.stashDepth;           // power of two
    const uint32_t ringMask = (uint32_t)args.ringSlots - 1u; // ringSlots: power of two

    // ---- which pixel is mine ----------------------------------------------------------------
    const uint32_t logicalBlock = (!KAT && args.blockOrder) ? args.blockOrder[blockIdx.x] : blockIdx.x;
    const int splitWave = SPLIT ? (int)(threadIdx.x >> 6) : 0;
    const int splitCount = SPLIT ? (int)(blockDim.x >> 6) : 1;
    const uint32_t slot = SPLIT ? logicalBlock * 64u + (uint32_t)lane : logicalBlock * blockDim.x + threadIdx.x; // index into the tile buffer
    // SPLIT: per-pass terms of the block, [nPasses][64] float4 behind the scene copy
    DFloat4* termTable = reinterpret_cast<DFloat4*>(ldsRaw + args.mailboxOffset);
    const int wave = (int)(slot >> 6);
    const int wavesPerTile = (args.tileW >> 3) * (args.tileH >> 3);
    const int ownedTile = wave / wavesPerTile;
    const int wb = wave - ownedTile * wavesPerTile;
    const int tile = args.tileIndex + ownedTile * args.tileCount;
    const int tx = tile % args.tilesX, ty = tile / args.tilesX;
    const int bxi = wb % (args.tileW >> 3), byi = wb / (args.tileW >> 3);
    const int px = tx * args.tileW + bxi * 8 + (lane & 7);
    const int py = ty * args.tileH + byi * 8 + (lane >> 3);
    const bool inImage = KAT ? (int)slot < args.katCount : (ownedTile < args.nTilesOwned && px < args.W && py < args.H);

    const int n = args.n;
    const uint32_t pixelIndex = (uint32_t)(py * args.W + px);
    // include/kajo_stream.h: key words (pixel, sample | pass << 16, seed lo ^ pass >> 16, seed hi) ^ constants
    const uint32_t keyA = pixelIndex ^ 0x61707865u;
    const uint32_t keyC = (uint32_t)args.seed ^ 0x79622d32u;
    const uint32_t keyD = (uint32_t)(args.seed >> 32) ^ 0x6b206574u;

    const F3 p1 = ld3(sc.p1), dp2 = ld3(sc.dp2), dp3 = ld3(sc.dp3), origin = ld3(sc.origin);
    const F3 background = ld3(sc.background);
    // x * pixelWidth and (H - y) * pixelHeight of Renderer.cpp:56-57 are constants of the pixel
    const float pixX = px * args.pixelWidth;
    const float pixY = (args.H - py) * args.pixelHeight;
    // The pixel whose pass the lane is ISSUING camera paths for: its own, or -- near the end of the wave's
    // life -- one taken over from a lane that still has whole passes left (see "pass stealing" below).
    uint32_t curKeyA = keyA;
    float curPixX = pixX, curPixY = pixY;

    // accumulated radiance of the pixel (Renderer.cpp:70-71), continued across launches
    // (the handle zeroes the buffer when it is created or reset)
    F3 total = f3(0.0f, 0.0f, 0.0f);
    float totalW = 0.0f;
    if (!KAT && inImage) {
        const float4 t = reinterpret_cast<const float4*>(args.tiles)[slot];
        total = f3(t.x, t.y, t.z);
        totalW = t.w;
    }
    bool katStarted = false;

    // ---- the lane's passes --------------------------------------------------------------------
    int mode = inImage ? DM_FREE : DM_DONE;
    const int passesMine = SPLIT ? args.nPasses / splitCount : args.nPasses; // the host launches SPLIT only when this divides
    const int firstMine = args.firstPass + splitWave * passesMine;
    const int lastPass = firstMine + passesMine; // exclusive
    // Pass stealing. A pass of a pixel is a self-contained piece of work (its n*n paths have their own
    // streams, its sum enters the pixel's total as one term), so a lane that has finished its own pixel
    // takes over the LAST not-yet-started pass of a lane that still has several to go, renders it, and
    // leaves radiance / S in a mailbox in LDS; the owner adds the mailbox terms after its own passes, in
    // pass order -- the float sums are formed exactly as without stealing. Only the last `stealWindow`
    // passes of a launch can be given away (that is all the imbalance there is, and bounds the mailbox).
    const int stealBase = lastPass - stealWindow > firstMine ? lastPass - stealWindow : firstMine;
    // ISSUE side: the pass whose camera paths are being started
    int pass = firstMine;            // (own or taken over)
    int ownPass = firstMine;         // next pass of the lane's own pixel that has not been opened
    int myEnd = inImage ? lastPass : firstMine; // own passes [.., myEnd); shrinks when one is taken over
    int stolenFrom = -1;             // lane whose pass is being issued, or -1
    int sampleX = 0, sampleY = n;    // sampleY == n: every sample of the issue pass has been started (or no pass is open yet)
    bool exhausted = false;          // no pass left to open, neither own nor anyone else's
    // RETIRE side: the pass whose samples are entering `radiance`. It is the issue pass itself (aheadBy == 0) or the one
    // before it (aheadBy == 1, described by aPass / aStolen); a lane never opens a pass while it is one ahead.
    F3 radiance = f3(0.0f, 0.0f, 0.0f); // sum over the retired samples of the retiring pass, in sample order
    int aLeft = 0;                   // samples of the retiring pass not yet retired (0: nothing open on the retire side)
    int aPass = 0, aStolen = -1, aheadBy = 0;
    uint32_t issued = 0, retired = 0; // sequence numbers of the lane's paths (mod 2^32; their difference is what matters)
    uint32_t doneMask = 0;            // ring slots that hold the radiance of a completed, not yet retired path
    int stashHead = 0, stashCount = 0; // FIFO of parked vertices: entries (stashHead + i) & (stashDepth - 1), i < stashCount

    // ---- the path in the lane's registers ------------------------------------------------------
    Rng rng{0, 0};
    F3 O = origin, d = f3(0.0f, 0.0f, 1.0f);
    F3 L = f3(0.0f, 0.0f, 0.0f), T = f3(1.0f, 1.0f, 1.0f);
    int depth = 0;
    bool collectEmission = true;
    uint32_t seq = 0;                // sequence number of the path
    // the vertex the path left last: position, object, path-weight scale (MIS correction of the extension ray, Shader.cpp:203-212)
    F3 vP = origin;
    int vId = 0;
    float vS = 0.0f;
    // extension ray sampled from the BSDF: weight pieces that wait for the light pdf of the hit
    bool pendBsdf = false;
    F3 pendF = L, pendT = L;
    float pendCos = 0.0f, pendP = 0.0f;
    // shadow ray in flight
    F3 pendContrib = L; // the light sample's contribution if the shadow ray reaches the light
    int lightObj = 0;   // object id of that light
    // DM_SHADOW_LAST: what the vertex adds once the shadow ray is back, and the extension ray that follows
    F3 vE = L, vLd = L, d2 = L;
    bool extOk = false;

    unsigned long long ctrTraversals = 0, ctrVertices = 0, ctrSlots = 0;
    const bool counting = args.counters != nullptr;
#ifdef KAJO_PROFILE
    // block profile: prof[2k] = wave executions of block k, prof[2k+1] = lanes active in it
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stampSum[5] = {0, 0, 0, 0, 0};
    unsigned long long stampLast = 0;
#define KAJO_STAMP(k)                                                                                                  \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        stampSum[k] += now_ - stampLast;                                                                               \
        stampLast = now_;                                                                                              \
    } while (0)
#define KAJO_PROF(k, cond)                                                                                             \
    do {                                                                                                               \
        const unsigned long long m_ = __ballot(cond);                                                                  \
        if (m_) {                                                                                                      \
            prof[2 * (k)] += 1;                                                                                        \
            prof[2 * (k) + 1] += __builtin_popcountll(m_);                                                             \
        }                                                                                                              \
    } while (0)
#else
#define KAJO_PROF(k, cond)                                                                                             \
    do {                                                                                                               \
    } while (0)
#define KAJO_STAMP(k)                                                                                                  \
    do {                                                                                                               \
    } while (0)
#endif

#ifdef KAJO_PROFILE
    stampLast = __builtin_amdgcn_s_memtime();
#endif
#if !KAJO_STRICT
    const float invS = krcp(args.S);
#endif
    const int nn = n * n;
    // the last light a vertex with object id `id` samples (a light does not sample itself, Shader.cpp:60-61): -1 if none
    const int lastLightAll = sc.nLights - 1;
    const int lastLightObj = sc.nLights > 0 ? np + 1 + lds.light[sc.nLights - 1] : -1;

    uint32_t trips = 0;
    for (;;) {
        trips++;
        KAJO_STAMP(4); // tail of the previous trip (retirement, loop back-edge)

        // ---- do the L blocks run in this trip? (wave-uniform) --------------------------------------
        const bool windowOpen = issued - retired <= ringMask + 1u; // ringSlots paths may wait in the ring behind the oldest
        const bool parked = mode == DM_FREE && stashCount > 0;
        const bool stuck = parked && (stashCount >= stashDepth || !windowOpen || exhausted || (sampleY == n && aheadBy != 0));
        bool runL = __builtin_popcountll(__ballot(parked)) >= args.thrL || __builtin_popcountll(__ballot(stuck)) >= args.thrStall;

        // ---- E work, before the ray: the camera ray of the pixel's next sample (Renderer.cpp:51-64) ---
        if (KAT && mode == DM_FREE && !(runL && parked)) {
            if (katStarted) {
                if (stashCount == 0)
                    mode = DM_DONE;
            } else {
                katStarted = true;
                O = ld3(args.katRays + 6 * slot);
                d = ld3(args.katRays + 6 * slot + 3);
                rng.lo = args.katStates[2 * slot];
                rng.hi = args.katStates[2 * slot + 1];
                L = f3(0.0f, 0.0f, 0.0f);
                T = f3(1.0f, 1.0f, 1.0f);
                depth = 0;
                collectEmission = true;
                pendBsdf = false;
                seq = issued++;
                mode = DM_EXTEND;
            }
        }
        KAJO_PROF(0, !KAT && mode == DM_FREE && !(runL && parked));
        if (!KAT && mode == DM_FREE && !(runL && parked)) {
            // (a) the issue pass is used up: open the next one -- an own pass, or one taken over
            bool wantPass = sampleY == n && aheadBy == 0 && !exhausted;
            bool opened = false;
            int newPass = 0, newStolen = -1;
            uint32_t newKey = keyA;
            float newX = pixX, newY = pixY;
            if (wantPass && ownPass < myEnd) {
                newPass = ownPass++;
                opened = true;
            }
            // out of own passes: take one over, or give up when nobody has one to give
            unsigned long long idleMask = __ballot(wantPass && !opened);
            while (idleMask) { // wave-uniform; only in the last stretch of the wave's life
                const int give = myEnd - 1; // the pass this lane could give away: its last, if not opened yet
                const unsigned long long giverMask = __ballot(!exhausted && give >= ownPass && give >= stealBase);
                if (giverMask == 0ull) {
                    if (wantPass && !opened)
                        exhausted = true;
                    break;
                }
                // lowest idle lane takes the last pass of the lowest giver
                const int thief = __builtin_ctzll(idleMask), giver = __builtin_ctzll(giverMask);
                const int takenPass = __builtin_amdgcn_readlane(myEnd, giver) - 1;
                const uint32_t gKey = (uint32_t)__builtin_amdgcn_readlane((int)keyA, giver);
                const float gX = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pixX), giver));
                const float gY = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pixY), giver));
                if (lane == giver)
                    myEnd = takenPass;
                if (lane == thief) {
                    opened = true;
                    newStolen = giver;
                    newPass = takenPass;
                    newKey = gKey;
                    newX = gX;
                    newY = gY;
                }
                idleMask &= idleMask - 1; // next idle lane
            }
            if (opened) {
                if (aLeft > 0) { // samples of the pass just issued are still in flight: it stays the retiring pass
                    aPass = pass;
                    aStolen = stolenFrom;
                    aheadBy = 1;
                } else {
                    aLeft = nn;
                }
                pass = newPass;
                stolenFrom = newStolen;
                curKeyA = newKey;
                curPixX = newX;
                curPixY = newY;
                sampleX = 0;
                sampleY = 0;
            }
            // (b) one camera path, if the lane may start one: a slot of the stash must be free for the vertex it may park
            if (sampleY < n && stashCount < stashDepth && windowOpen) {
                uint32_t a = curKeyA, c = keyC ^ ((uint32_t)pass >> 16), dd = keyD;
                uint32_t b = ((uint32_t)(sampleY * n + sampleX) | ((uint32_t)pass << 16)) ^ 0x3320646eu;
                KAJO_QUARTER_ROUND(a, b, c, dd);
                KAJO_QUARTER_ROUND(a, b, c, dd);
                KAJO_QUARTER_ROUND(a, b, c, dd);
                rng.lo = (uint64_t)a | ((uint64_t)b << 32);
                rng.hi = (uint64_t)c | ((uint64_t)dd << 32);
                rngStep(rng);
                float offX = unitBits((uint32_t)rng.lo);
                float offY = unitBits((uint32_t)(rng.lo >> 32));
                float sx = curPixX + sampleX * args.sampleWidth + offX * args.sampleWidth;
                float sy = curPixY + sampleY * args.sampleHeight + offY * args.sampleHeight;
                F3 dir = p1 + dp2 * sx + dp3 * sy - origin;
                d = normalize(dir);
                O = origin;
                L = f3(0.0f, 0.0f, 0.0f);
                T = f3(1.0f, 1.0f, 1.0f);
                depth = 0;
                collectEmission = true;
                pendBsdf = false;
                seq = issued++;
                sampleX++;
                if (sampleX == n) {
                    sampleX = 0;
                    sampleY++;
                }
                mode = DM_EXTEND;
            } else if (exhausted && issued == retired) {
                mode = DM_DONE;
            }
        }
        KAJO_STAMP(0); // camera-ray block

        // ---- L work, before the ray: resume a parked vertex (Shader.cpp:50-86,180-200) --------------------
        // When no lane of the wave has a ray to trace, the parked vertices are all there is to do.
        if (!runL && __ballot(mode == DM_EXTEND || mode == DM_SHADOW_MID || mode == DM_SHADOW_LAST) == 0ull)
            runL = true;
        KAJO_PROF(1, runL && mode == DM_FREE && stashCount > 0);
        if (runL && mode == DM_FREE && stashCount > 0) {
            DFloat4* e = stash + (size_t)stashHead * (KAJO_STASH_QUADS * 64);
            const DFloat4 q0 = e[0], q1 = e[64], q2 = e[128], q3 = e[192], q4 = e[256], q5 = e[320];
            rng.lo = (uint64_t)__builtin_bit_cast(uint32_t, q0.x) | ((uint64_t)__builtin_bit_cast(uint32_t, q0.y) << 32);
            rng.hi = (uint64_t)__builtin_bit_cast(uint32_t, q0.z) | ((uint64_t)__builtin_bit_cast(uint32_t, q0.w) << 32);
            L = f3(q1.x, q1.y, q1.z);
            T = f3(q2.x, q2.y, q2.z);
            vP = f3(q3.x, q3.y, q3.z);
            const F3 vN = f3(q4.x, q4.y, q4.z);
            const F3 view = f3(q5.x, q5.y, q5.z);
            vLd = f3(q3.w, q4.w, q5.w);
            const uint32_t pk0 = __builtin_bit_cast(uint32_t, q1.w), pk1 = __builtin_bit_cast(uint32_t, q2.w);
            vId = (int)(pk0 & 0x3ffffu);
            depth = (int)((pk0 >> 18) & 0x3ffu);
            collectEmission = ((pk0 >> 28) & 1u) != 0u;
            const int vKind = (int)((pk0 >> 29) & 3u);
            int lightK = (int)(pk1 & 0xffffu);
            seq = (pk1 >> 16) & 0xffu; // sequence numbers are compared in their low 8 bits: a lane has a handful of paths in flight
            const DMaterial& m = lds.material[vId - 1];
            vE = collectEmission ? ld3(m.emission) : f3(0.0f, 0.0f, 0.0f); // Shader.cpp:121
            const F3 vColor = vKind == 0 ? ld3(m.diffuse) : ld3(m.specular);
            const float vExp = m.exponent;
            vS = vKind == 0 ? m.sDiffuse : m.sSpecular; // 1/pc * 1/pt * 1/pd, Shader.cpp:160-177 (formed on the host in this order)
            const F3 vR = reflect(view, vN);
            const int lastLight = vId == lastLightObj ? lastLightAll - 1 : lastLightAll;

            // ---- sampleLights (Shader.cpp:50-86), one light per visit ------------------------------
            bool shadowRay = false;
            while (lightK < sc.nLights) {
                // Lights whose sample is discarded whatever it is -- the ideal reflector asks for none (its pdf toward any
                // given direction is 0, BSDF.cpp:93-96), and a light that lies wholly below the vertex's horizon has
                // max(0, n.l) = 0 for every point of it -- only draw their random number (Light.cpp:39-41: one draw per
                // sample), in a loop of its own: the full sampling code below then runs for lights that can count. Scenes
                // of many lights only (16 lights: +6 % FAST, +25 % STRICT); with one light the test is pure overhead (-4.6 %).
                if (!COLD_LDS && sc.nLights >= 4) { // (large-scene kernels only: in the small-scene loop the extra code costs 5 % by itself)
                    for (; lightK < sc.nLights; lightK++) {
                        const int sk = lds.light[lightK];
                        if (np + 1 + sk == vId) // a light does not sample itself (and draws nothing)
                            continue;
                        const DSphereCold& lk = lds.lightCold[lightK];
                        const F3 toC = f3(lk.cx - vP.x, lk.cy - vP.y, lk.cz - vP.z);
                        const bool below = dot(vN, toC) < -(1.001f * lk.radius + 1e-6f * (__builtin_fabsf(toC.x) + __builtin_fabsf(toC.y) + __builtin_fabsf(toC.z)));
                        if (!(vKind == 2 || below))
                            break;
                        rngStep(rng);
                    }
                    if (lightK >= sc.nLights)
                        break;
                }
                const int si = lds.light[lightK];
                if (np + 1 + si == vId) { // a light does not sample itself
                    lightK++;
                    continue;
                }
                const DSphereCold& lc = lds.lightCold[lightK];
                float pl;
                // written straight into the ray: a discarded sample leaves d and O to the next light or to the BSDF sample
                d = lightGenerate(f3(lc.cx, lc.cy, lc.cz), lc.radius, vP, rng, pl);
                const F3 l = d;
#if !KAJO_STRICT
                pl = lightPdf(lc, vP);
#endif
                // The reference traces first and asks the BSDF afterwards; a zero BSDF pdf (always for
                // the reflector, outside the lobe for Phong) discards the sample either way, so the
                // trace is skipped for it.
                float pb;
                const F3 fl = bsdfEvaluateWithPdf(vKind, vColor, vExp, vR, vN, l, pb);
                // A light at or below the horizon contributes max(0, n.l) = 0 whatever the shadow ray
                // finds (the sum stays as it is: x + (+-0) == x), so that walk is skipped as well.
                const float cosL = kmax0(dot(vN, l));
                O = vP + l * kEps;
                if (pl == 0.0f || pb == 0.0f || cosL == 0.0f) {
                    lightK++;
                    continue;
                }
                const DFloat4 le = lds.lightEmission[lightK];
                const F3 Le = f3(le.x, le.y, le.z);
                pendContrib = ((krcp(pb + pl) * fl) * cosL) * Le;
                lightObj = np + 1 + lds.light[lightK];
                shadowRay = true;
                break;
            }
            KAJO_PROF(2, shadowRay);
            if (shadowRay && lightK < lastLight) {
                // more lights follow: the vertex stays parked at the head of the FIFO with the stream and the light index
                // moved on; the shadow ray's answer is added to its vLd after the trace
                e[0] = DFloat4{__builtin_bit_cast(float, (uint32_t)rng.lo), __builtin_bit_cast(float, (uint32_t)(rng.lo >> 32)),
                               __builtin_bit_cast(float, (uint32_t)rng.hi), __builtin_bit_cast(float, (uint32_t)(rng.hi >> 32))};
                reinterpret_cast<uint32_t*>(e + 128)[3] = (pk1 & 0xffff0000u) | (uint32_t)(lightK + 1);
                mode = DM_SHADOW_MID;
            } else {
                // ---- BSDF sampling (Shader.cpp:191-200): the vertex leaves the stash ------------------
                KAJO_PROF(3, true);
                stashHead = (stashHead + 1) & (stashDepth - 1);
                stashCount--;
                F3 tg = f3(0.0f, 0.0f, 0.0f), bn = tg;
                if (vKind == 0) {
                    if (vId <= np) {
                        const DFloat4 t4 = lds.planeFrame[3 * (vId - 1) + 1], b4 = lds.planeFrame[3 * (vId - 1) + 2];
                        tg = f3(t4.x, t4.y, t4.z);
                        bn = f3(b4.x, b4.y, b4.z);
                    } else {
                        sphereFrame(vN, tg, bn);
                    }
                }
                float p;
                F3 fd;
                d2 = bsdfGenerate(vKind, vColor, vExp, vR, vN, tg, bn, rng, p, fd);
                pendF = fd;
                pendCos = kmax0(dot(vN, d2));
                pendP = p;
                pendBsdf = true;
                pendT = T;
                T = T * (vS * ((krcp(0.0f + p) * pendF) * pendCos));
                depth++;
                extOk = p != 0.0f;
                if (shadowRay) {
                    mode = DM_SHADOW_LAST; // (O, d) is the shadow ray; L waits for its answer
                } else {
                    L = L + pendT * (vS * (vE + vLd));
                    O = vP + d2 * kEps;
                    d = d2;
                    mode = extOk ? DM_EXTEND : DM_RETIRE;
                }
            }
        }
        KAJO_STAMP(1); // light + BSDF sampling block

        {
            const unsigned long long aliveMask = __ballot(mode != DM_DONE);
            if (aliveMask == 0ull)
                break;
        }
        const bool hasRay = mode == DM_EXTEND || mode == DM_SHADOW_MID || mode == DM_SHADOW_LAST;

        // ---- one ray per lane through the whole scene ------------------------------------------
        const Hit hit = trace<!COLD_LDS>(sc, lds, O, d);
        KAJO_STAMP(2); // traversal
        if (counting) {
            ctrTraversals += __builtin_popcountll(__ballot(hasRay));
            ctrSlots += 64;
        }

        // ---- L work, after the ray: the shadow ray's answer -------------------------------------------
        KAJO_PROF(7, mode == DM_SHADOW_MID || mode == DM_SHADOW_LAST);
        if (mode == DM_SHADOW_MID) {
            // Shader.cpp:72-73: the sample counts iff the closest hit of the shadow ray IS the light
            if (hit.id == lightObj) {
                float* e = reinterpret_cast<float*>(stash + (size_t)stashHead * (KAJO_STASH_QUADS * 64));
                e[192 * 4 + 3] = e[192 * 4 + 3] + pendContrib.x;
                e[256 * 4 + 3] = e[256 * 4 + 3] + pendContrib.y;
                e[320 * 4 + 3] = e[320 * 4 + 3] + pendContrib.z;
            }
            mode = DM_FREE;
        } else if (mode == DM_SHADOW_LAST) {
            if (hit.id == lightObj)
                vLd = vLd + pendContrib;
            L = L + pendT * (vS * (vE + vLd));
            O = vP + d2 * kEps;
            d = d2;
            mode = extOk ? DM_EXTEND : DM_RETIRE;
        } else if (mode == DM_EXTEND) {
            // ---- E work, after the ray: the vertex (Shader.cpp:113-178) -----------------------------------
            const DFloat4* mq = reinterpret_cast<const DFloat4*>(lds.material + (hit.id > 0 ? hit.id - 1 : 0));
            const DFloat4 m0 = mq[0], m1 = mq[1];
            const uint32_t m1flags = __builtin_bit_cast(uint32_t, m1.w);
            // Weight of the BSDF-sampled segment that just ended (Shader.cpp:203-212). The throughput was
            // advanced with a zero light pdf when the direction was sampled (0 + p == p exactly); only a ray
            // that lands on a light other than the vertex it left needs the MIS denominator pL + p.
            if (pendBsdf) {
                // (A light that is a pure emitter -- no diffuse, specular or transparent colour, pRR == 0 -- ends every path
                // that reaches it, and a path that arrives over a BSDF-sampled segment collects no emission there
                // (Shader.cpp:121,212): its throughput is never used again, so the MIS correction is skipped. STRICT
                // keeps it when the throughput is not finite: NaN * 0 must stay NaN.)
#if KAJO_STRICT
                const bool weightMatters = m0.x != 0.0f || !(__builtin_fabsf(T.x) < __builtin_inff() && __builtin_fabsf(T.y) < __builtin_inff() && __builtin_fabsf(T.z) < __builtin_inff());
#else
                const bool weightMatters = m0.x != 0.0f;
#endif
                if (hit.id > np && hit.id != vId && (m1flags & KAJO_MAT_IS_LIGHT) && weightMatters) {
                    const DSphereCold& lc = lds.sphereCold[hit.id - 1 - np];
#if KAJO_STRICT
                    const float pL = krcp(solidAngle(f3(lc.cx, lc.cy, lc.cz), lc.radius, vP));
#else
                    const float pL = lightPdf(lc, vP);
#endif
                    const F3 wb = (krcp(pL + pendP) * pendF) * pendCos;
                    T = pendT * (vS * wb);
                }
                collectEmission = false; // SampleNonEmissiveObjects
                pendBsdf = false;
            }
            KAJO_PROF(4, hit.id != 0);
            if (hit.id == 0) { // Shader.cpp:116-117
                L = L + T * background;
                mode = DM_RETIRE;
            } else {
                if (counting)
                    ctrVertices += 1;
                const F3 view = d;
                vP = O + d * hit.t; // Raytracer.cpp:134-135
                const F3 vN = hitNormal(sc, lds, hit, O, d);
                vId = hit.id;
                const F3 em = collectEmission ? f3(m1.x, m1.y, m1.z) : f3(0.0f, 0.0f, 0.0f); // Shader.cpp:121
                float pc;
                const bool cont = flipCoin(rng, m0.x, pc); // Shader.cpp:124-125
                if (!cont || depth >= args.depthLimit) {
                    // Shader.cpp:126-127: 1 / pc with pc = pRR (depth limit) or 1 - pRR (the coin said stop), formed on the host
                    float sEnd = m0.w;
                    if (cont)
                        sEnd = mq[4].w;
                    L = L + T * (sEnd * em);
                    mode = DM_RETIRE;
                } else {
                    float pt;
                    const bool transparent = flipCoin(rng, m0.y, pt); // Shader.cpp:130-134
                    KAJO_PROF(5, transparent);
                    if (transparent) { // Shader.cpp:137-151; the BSDF colour is the SPECULAR colour
                        const DFloat4 m2 = mq[2], m4 = mq[4];
                        F3 nd = transmissionDirection(view, vN, m2.w);
                        float cosA = __builtin_fabsf(dot(nd, vN));
                        F3 spec = f3(m2.x, m2.y, m2.z);
                        F3 f = f3(kdiv(spec.x, cosA), kdiv(spec.y, cosA), kdiv(spec.z, cosA)); // BSDF.cpp:126-130
                        F3 w = (m4.z * f) * __builtin_fabsf(dot(vN, nd)); // sTransparent = 1/pc * 1/pt
                        L = L + T * (w * em);
                        T = T * w;
                        O = vP + nd * kEps;
                        d = nd;
                        depth++;
                        // mode stays DM_EXTEND, the light sampling scheme is inherited
                    } else {
                        // The vertex survives and wants its lights and a BSDF sample: park it (the lane had a free slot
                        // when it traced this ray) and free the registers for the pixel's next camera path.
                        float pd;
                        const bool diffuse = flipCoin(rng, m0.z, pd); // Shader.cpp:153-154
                        const uint32_t vKind = diffuse ? 0u : ((m1flags & KAJO_MAT_HAS_EXPONENT) ? 1u : 2u);
                        KAJO_PROF(6, true);
                        DFloat4* e = stash + (size_t)((stashHead + stashCount) & (stashDepth - 1)) * (KAJO_STASH_QUADS * 64);
                        const uint32_t pk0 = (uint32_t)vId | ((uint32_t)depth << 18) | (collectEmission ? 1u << 28 : 0u) | (vKind << 29);
                        const uint32_t pk1 = (seq & 0xffu) << 16; // light index 0
                        e[0] = DFloat4{__builtin_bit_cast(float, (uint32_t)rng.lo), __builtin_bit_cast(float, (uint32_t)(rng.lo >> 32)),
                                       __builtin_bit_cast(float, (uint32_t)rng.hi), __builtin_bit_cast(float, (uint32_t)(rng.hi >> 32))};
                        e[64] = DFloat4{L.x, L.y, L.z, __builtin_bit_cast(float, pk0)};
                        e[128] = DFloat4{T.x, T.y, T.z, __builtin_bit_cast(float, pk1)};
                        e[192] = DFloat4{vP.x, vP.y, vP.z, 0.0f};
                        e[256] = DFloat4{vN.x, vN.y, vN.z, 0.0f};
                        e[320] = DFloat4{view.x, view.y, view.z, 0.0f};
                        stashCount++;
                        mode = DM_FREE;
                    }
                }
            }
        }
        KAJO_STAMP(3); // vertex / shadow-result block

        // ---- a completed path: its radiance joins the pixel's sum in sample order (Renderer.cpp:66) ----------
        if (mode == DM_RETIRE) {
            mode = DM_FREE;
            if (KAT) {
                reinterpret_cast<float4*>(args.katRgb)[slot] = make_float4(L.x, L.y, L.z, 0.0f);
                args.katFinal[2 * slot] = rng.lo;
                args.katFinal[2 * slot + 1] = rng.hi;
                retired++;
            } else if (((seq ^ retired) & 0xffu) != 0u) {
                // an older path of this lane is still parked: wait in the ring
                const uint32_t s = seq & ringMask;
                ring[(s * 3 + 0) * 64] = L.x;
                ring[(s * 3 + 1) * 64] = L.y;
                ring[(s * 3 + 2) * 64] = L.z;
                doneMask |= 1u << s;
            } else {
                F3 add = L;
                for (;;) {
                    radiance = radiance + add;
                    retired++;
                    if (--aLeft == 0) { // the retiring pass is complete: Renderer.cpp:70-71
#if KAJO_STRICT
                        const F3 term = f3(radiance.x / args.S, radiance.y / args.S, radiance.z / args.S);
#else
                        const F3 term = radiance * invS;
#endif
                        const int tPass = aheadBy ? aPass : pass, tStolen = aheadBy ? aStolen : stolenFrom;
                        if (SPLIT) // own or taken over: the term goes to the table, under its pass and pixel
                            termTable[(tPass - args.firstPass) * 64 + (tStolen >= 0 ? tStolen : lane)] = DFloat4{term.x, term.y, term.z, 0.0f};
                        else if (tStolen >= 0)
                            mailbox[tStolen * stealWindow + (tPass - stealBase)] = DFloat4{term.x, term.y, term.z, 0.0f};
                        else
                            total = total + term;
                        radiance = f3(0.0f, 0.0f, 0.0f);
                        if (aheadBy) { // the issue pass becomes the retiring pass; none of its samples has retired yet
                            aheadBy = 0;
                            aLeft = nn;
                        }
                    }
                    const uint32_t s = retired & ringMask;
                    if (!((doneMask >> s) & 1u))
                        break;
                    doneMask &= ~(1u << s);
                    add = f3(ring[(s * 3 + 0) * 64], ring[(s * 3 + 1) * 64], ring[(s * 3 + 2) * 64]);
                }
            }
        }
    }

    if (SPLIT) {
        __syncthreads(); // every wave of the block has left its loop: the table is complete
        if (splitWave == 0 && inImage) {
            for (int p = 0; p < args.nPasses; p++) { // Renderer.cpp:70-71, pass by pass
                const DFloat4 t = termTable[p * 64 + lane];
                total = total + f3(t.x, t.y, t.z);
            }
            reinterpret_cast<float4*>(args.tiles)[slot] = make_float4(total.x, total.y, total.z, totalW);
        }
    } else if (!KAT && inImage) {
        // passes of this pixel that other lanes rendered, in pass order
        for (int p = myEnd; p < lastPass; p++) {
            const DFloat4 t = mailbox[lane * stealWindow + (p - stealBase)];
            total = total + f3(t.x, t.y, t.z);
        }
        reinterpret_cast<float4*>(args.tiles)[slot] = make_float4(total.x, total.y, total.z, totalW);
    }
    if (!KAT && !SPLIT && args.waveTrips && lane == 0)
        args.waveTrips[slot >> 6] = trips;

    if (counting && lane == 0) {
        atomicAdd(&args.counters[0], ctrTraversals);
        atomicAdd(&args.counters[2], ctrSlots);
#ifdef KAJO_PROFILE
        for (int k = 0; k < 16; k++)
            atomicAdd(&args.counters[4 + k], prof[k]);
        for (int k = 0; k < 5; k++)
            atomicAdd(&args.counters[20 + k], stampSum[k]);
#endif
    }
    if (counting) {
        // vertices are per lane: reduce over the wave first
        unsigned long long v = ctrVertices;
        for (int o = 32; o > 0; o >>= 1)
            v += __shfl_down(v, o);
        if (lane == 0)
            atomicAdd(&args.counters[1], v);
    }
}

} // namespace

extern "C" __global__ void __launch_bounds__(256, KAJO_WAVES_PER_SIMD_BIG) KAJO_KERNEL_NAME_DEFERRED(const RenderArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    renderBodyDeferred<true, false>(args, ldsRaw);
}

extern "C" __global__ void __launch_bounds__(256, KAJO_WAVES_PER_SIMD_BIG) KAJO_KERNEL_NAME_DEFERRED_BIG(const RenderArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    renderBodyDeferred<false, false>(args, ldsRaw);
}

extern "C" __global__ void __launch_bounds__(256, KAJO_WAVES_PER_SIMD_BIG) KAJO_KAT_SHADE_NAME_DEFERRED(const RenderArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    renderBodyDeferred<false, true>(args, ldsRaw);
}
