"""Round 6: the miniature pipeline (streams.mini_pipeline_ms) over every (tokenizer stream, dense pair) of a process whose application used `junk`
normal + 2 high-priority streams first.  ratio = ms per step / ms of one dense graph alone: ~0.6 = the two dense streams overlap, >= 1 = they do not."""
import itertools
import sys
import torch

sys.path.insert(0, ".")
from point_sam_amd.streams import mini_pipeline_ms, streams_alias

junk = int(sys.argv[1])
torch.cuda.init()
x = torch.zeros(8, device="cuda")
for s in [torch.cuda.Stream() for _ in range(junk)] + [torch.cuda.Stream(priority=-1) for _ in range(2 if junk else 0)]:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
NN, NH = 8, 5
hs = [torch.cuda.Stream(priority=-1) for _ in range(NH)]
ns = [torch.cuda.Stream() for _ in range(NN)]
null = torch.cuda.default_stream()
for s in hs + ns:                       # first use, in this order
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
cls = []
for i, s in enumerate([null] + ns):     # alias classes of the normal streams (= hardware queues)
    for c in cls:
        if streams_alias(c[0][1], s):
            c.append((i, s)); break
    else:
        cls.append([(i, s)])
print("junk", junk, "normal queue classes:", [[("null" if i == 0 else f"n{i - 1}") for i, _ in c] for c in cls], flush=True)
hcls = []
for i, s in enumerate(hs):
    for c in hcls:
        if streams_alias(c[0][1], s):
            c.append((i, s)); break
    else:
        hcls.append([(i, s)])
print("high queue classes:", [[f"h{i}" for i, _ in c] for c in hcls], flush=True)
reps = [c[0] for c in cls if c[0][0] != 0]      # one stream per normal queue other than the default stream's
null_mates = [m for c in cls if c[0][0] == 0 for m in c[1:]]
if null_mates:
    reps.append(null_mates[0])
for hi, h in [c[0] for c in hcls]:
    row = []
    for (i, a), (j, b) in itertools.combinations(reps, 2):
        per, alone = mini_pipeline_ms(h, [a, b], steps=8)
        row.append(f"n{i - 1}+n{j - 1}: {per / alone:4.2f}")
    print(f"tok h{hi} | " + "  ".join(row), flush=True)
