"""How many pixels survive candidate FAST pre-tests, and how many are real corners (oracle pyramid of one synthetic 752x480 frame)?
Decides whether a stronger pre-filter in front of the exact score can pay (round 2: it cannot -- 57 % of the 4-point survivors are corners)."""
import sys, numpy as np, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import orbref
synth = importlib.import_module("orb-slam3_amd.synth")
kind = sys.argv[1] if len(sys.argv) > 1 else 'textured'
img = synth.gen_image(752,480,1,kind)
ref = orbref.Extractor(1000)
ref(img,(0,1000))
ring = [(0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),(0,-3),(-1,-3),(-2,-2),(-3,-1),(-3,0),(-3,1),(-2,2),(-1,3)]
tot = dict(px=0,q4=0,q8=0,q8b=0,corner=0,qrow=0)
for l in range(8):
    L = ref.level_image(l).astype(np.int32)
    h,w = L.shape
    v = L[3:h-3,3:w-3]
    R = np.stack([L[3+dy:h-3+dy, 3+dx:w-3+dx] for dx,dy in ring])   # [16,h-6,w-6]
    t = 20
    dark = R < v - t; bright = R > v + t
    def arc9(m):
        ok = np.zeros(m.shape[1:],bool)
        for s in range(16):
            a = np.ones(m.shape[1:],bool)
            for k in range(9): a &= m[(s+k)%16]
            ok |= a
        return ok
    corner = arc9(dark)|arc9(bright)
    q4 = ((dark[0]|dark[8])&(dark[4]|dark[12])) | ((bright[0]|bright[8])&(bright[4]|bright[12]))
    def q8f(m):
        T = m[7]&m[8]&m[9]; B = m[15]&m[0]&m[1]
        return (T|B)&(m[4]|m[12])
    q8 = q8f(dark)|q8f(bright)
    # variant: also require left/right triples? rows y-1,y+1 not loaded. alternative q8b: top3|bot3 AND (l|r), plus 4-point both
    def q8g(m):   # even positions: 4 consecutive even positions
        e=[m[2*j] for j in range(8)]
        ok=np.zeros(m.shape[1:],bool)
        for j in range(8): ok |= e[j]&e[(j+1)%8]&e[(j+2)%8]&e[(j+3)%8]
        return ok
    q8b = q8g(dark)|q8g(bright)
    n=v.size
    print("L%d px %7d  q4 %.3f  q8(rows) %.3f  q8(even) %.3f  corner %.3f   corner/q4 %.2f corner/q8 %.2f"%(l,n,q4.mean(),q8.mean(),q8b.mean(),corner.mean(),corner.sum()/q4.sum(),corner.sum()/q8.sum()))
    #assert not (corner & ~q4).any() and not (corner & ~q8).any() and not (corner&~q8b).any()
    tot['px']+=n; tot['q4']+=q4.sum(); tot['q8']+=q8.sum(); tot['q8b']+=q8b.sum(); tot['corner']+=corner.sum()
print({k:(v/tot['px'] if k!='px' else v) for k,v in tot.items()})

# ---- polarity of the 4-point survivors: dark-only / bright-only / both (a single-polarity score network would need two passes for "both")
tot2 = dict(q4=0, dark_only=0, bright_only=0, both=0)
for l in range(8):
    L = ref.level_image(l).astype(np.int32)
    h, w = L.shape
    v = L[3:h-3, 3:w-3]
    R = {k: L[3+dy:h-3+dy, 3+dx:w-3+dx] for k, (dx, dy) in enumerate(ring) if k in (0, 4, 8, 12)}
    t = 20
    D = ((R[0] < v - t) | (R[8] < v - t)) & ((R[4] < v - t) | (R[12] < v - t))
    B = ((R[0] > v + t) | (R[8] > v + t)) & ((R[4] > v + t) | (R[12] > v + t))
    tot2['q4'] += int((D | B).sum()); tot2['dark_only'] += int((D & ~B).sum()); tot2['bright_only'] += int((B & ~D).sum()); tot2['both'] += int((D & B).sum())
print("polarity of 4-point survivors:", {k: round(v / tot2['q4'], 4) for k, v in tot2.items()})
