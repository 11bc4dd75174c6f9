"""Brute-force bank-conflict count of a ds_read_b128 fragment read on MI355X (lane groups and bank rule: MI355X_MICROARCH.md).
Usage inside a session: python tools/lds_conflicts.py  - searches XOR slot swizzles for 64-byte (4-slot) and 128-byte (8-slot) rows
read by lanes (li = lane & 15 -> row li + shift, g = lane >> 4 -> K chunk g)."""
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]

def cycles(addr_of_lane):
    """LDS cycles of one ds_read_b128: per lane group, max over banks of distinct 16-byte slots on that bank"""
    tot = 0
    for grp in GROUPS:
        banks = {}
        for l in grp:
            a = addr_of_lane(l)
            for d in range(4):
                banks.setdefault(((a // 4) + d) % 64, set()).add(a // 16)
        tot += max(len(v) for v in banks.values())
    return tot

def frag_read(row_bytes, swz, shift, kchunks_per_read=4):
    slots = row_bytes // 16
    def addr(l):
        li, g = l & 15, l >> 4
        row = li + shift
        return row * row_bytes + ((g ^ swz(row)) % slots) * 16
    return cycles(addr)

if __name__ == "__main__":
    for row_bytes in (64, 128):
        slots = row_bytes // 16
        print(f"rows of {row_bytes} bytes ({slots} slots); ideal = 4 cycles per read")
        cands = {}
        for sh in range(0, 4):
            for msk in range(1, slots):
                cands[f"(row>>{sh})&{msk}"] = (lambda r, sh=sh, msk=msk: (r >> sh) & msk)
        for mul in (1, 2, 3):
            cands[f"(row*{mul}>>1)&{slots-1}"] = (lambda r, mul=mul: (r * mul >> 1) & (slots - 1))
        for name, f in cands.items():
            res = [frag_read(row_bytes, f, s) for s in range(3)]
            if max(res) <= 4: print("  conflict-free for shifts 0,1,2:", name, res)
        best = sorted(((max(frag_read(row_bytes, f, s) for s in range(3)), n) for n, f in cands.items()))[:5]
        print("  best:", best)
