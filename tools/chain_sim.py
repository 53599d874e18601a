"""List-scheduling simulator for the persistent DiT forward (csrc/chain.hip): given per-phase item times (tools/chain_spans.py) it replays
the in-order claiming of a static item list by 256 workgroups with sample-granular dependencies, and derives better lists:
  order 0  phase-major (block, phase, sample, sub) -- what the first version shipped
  order 1  the order in which a GREEDY scheduler (a free workgroup takes the ready item of the sample that is furthest behind) started
           the items -- every dependency still precedes its dependents, so the list stays deadlock-free
    python tools/chain_sim.py [B] [depth]"""
import heapq
import sys

PH = ["qkv", "attn", "proj", "ln2", "fc1", "fc2", "red"]


def phases(B, heads=16, tn_qkv=14, tn_d=5, tn_fc1=18, S=3, ln_items=16):
    return [tn_qkv, heads, tn_d, ln_items, tn_fc1, tn_d * S, ln_items]


def phase_major(B, depth, per):
    items = []
    for blk in range(depth):
        for ph in range(7):
            for g in range(B):
                for t in range(per[ph]):
                    items.append((blk, ph, g, t))
    return items


def replay(items, per, dur, W=256, claim=1.5):
    """in-order claims; returns (makespan, busy fraction)"""
    nph = 7
    done_cnt = {}
    ready_at = {}          # (g, global phase index) -> time all its items finished
    free = [(0.0, w) for w in range(W)]
    heapq.heapify(free)
    fin = {}
    busy = 0.0
    for (blk, ph, g, t) in items:
        tw, w = heapq.heappop(free)
        gp = blk * nph + ph
        dep = ready_at.get((g, gp - 1), 0.0) if gp > 0 else 0.0
        if gp > 0 and (g, gp - 1) not in ready_at:
            raise RuntimeError("dependency behind its dependent")
        start = max(tw + claim, dep)
        end = start + dur[ph]
        busy += dur[ph]
        k = (g, gp)
        done_cnt[k] = done_cnt.get(k, 0) + 1
        fin[k] = max(fin.get(k, 0.0), end)
        if done_cnt[k] == per[ph]:
            ready_at[k] = fin[k]
        heapq.heappush(free, (end, w))
    span = max(t for t, _ in free)
    return span, busy / (W * span)


def greedy(B, depth, per, dur, W=256, claim=1.5, prio="behind"):
    """event simulation of a dynamic scheduler; returns the start order as an item list"""
    nph = 7
    total_ph = depth * nph
    nxt = [0] * B                  # per sample: current global phase
    issued = [0] * B               # items of the current phase already started
    finished = [0] * B
    order = []
    events = []                    # (time, kind, payload)
    free_w = list(range(W))
    t = 0.0
    running = 0

    def ready_samples():
        return [g for g in range(B) if nxt[g] < total_ph and issued[g] < per[nxt[g] % nph]]
    while True:
        rs = ready_samples()
        while free_w and rs:
            if prio == "behind":
                g = min(rs, key=lambda s: (nxt[s], issued[s]))
            else:
                g = max(rs, key=lambda s: (dur[nxt[s] % nph], -nxt[s]))
            gp = nxt[g]
            order.append((gp // nph, gp % nph, g, issued[g]))
            issued[g] += 1
            w = free_w.pop()
            heapq.heappush(events, (t + claim + dur[gp % nph], w, g, gp))
            rs = ready_samples()
        if not events:
            break
        t, w, g, gp = heapq.heappop(events)
        free_w.append(w)
        finished[g] += 1
        if finished[g] == per[gp % nph] and nxt[g] == gp:
            nxt[g] += 1
            issued[g] = 0
            finished[g] = 0
    return order, t


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 28
    per = phases(B)
    for label, dur in (("measured v5", [83.6, 48.1, 83.6, 12.4, 92.5, 109.6, 31.4]), ("target", [80, 32, 80, 7, 90, 105, 12])):
        pm = phase_major(B, depth, per)
        span, busy = replay(pm, per, dur)
        print(f"{label}: phase-major  {span / 1000:.2f} ms  busy {busy:.3f}  (sum of bodies / 256 = {sum(d * p for d, p in zip(dur, per)) * B * depth / 256 / 1000:.2f} ms)")
        for prio in ("behind", "long"):
            od, tg = greedy(B, depth, per, dur, prio=prio)
            span, busy = replay(od, per, dur)
            print(f"{label}: greedy[{prio}]  dynamic {tg / 1000:.2f} ms; its start order replayed in order {span / 1000:.2f} ms  busy {busy:.3f}")
