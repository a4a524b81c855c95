"""dev: topology of the captured training-step hipGraph (hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType through ctypes):
node kinds, forks / joins, and for every MEMSET node what precedes and follows it.  usage: [MPHIP_GRAPH_MEMSET_FIX=0] dbg_graph_topology.py [mse|staged]   (with the fix on, the memset nodes are already kernels)"""
import sys, os, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import hotpath_ref as R
from megaportrait_hack_amd import model as M, training
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "mse"
sd = R.seeded_state_dict(R.g3d_shapes(96), 91, prefix="G3d.")
x = R.seeded_tensor((1, 96, 8, 16, 16), 92).to(dev)
g = M.G3d(96); g.load_state_dict({k[len("G3d."):]: v for k, v in sd.items()}); g = g.to(dev).train()
tgt = R.seeded_tensor((1, 96, 8, 16, 16), 93).to(dev)
def loss_fn(m, x):
    y = m(x)
    if kind == "mse": return F.mse_loss(y, tgt)
    return (y - tgt).square().view(-1, 1024).sum(1).sum() / y.numel()
opt = torch.optim.SGD(g.parameters(), lr=1e-3)
step = training.GraphedTrainStep(g, loss_fn, opt, {"x": x}, warmup=2)
hip = ctypes.CDLL("libamdhip64.so")
graph = ctypes.c_void_p(step.graph.raw_cuda_graph())
n = ctypes.c_size_t(0)
assert hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) == 0
nodes = (ctypes.c_void_p * n.value)()
assert hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) == 0
ne = ctypes.c_size_t(0)
assert hip.hipGraphGetEdges(graph, None, None, ctypes.byref(ne)) == 0
fr, to = (ctypes.c_void_p * ne.value)(), (ctypes.c_void_p * ne.value)()
assert hip.hipGraphGetEdges(graph, fr, to, ctypes.byref(ne)) == 0
TYPES = {0: "KERNEL", 1: "MEMCPY", 2: "MEMSET", 3: "HOST", 4: "GRAPH", 5: "EMPTY", 6: "WAIT_EVENT", 7: "EVENT_RECORD"}
typ = {}
for nd in nodes:
    t = ctypes.c_int(-1)
    hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
    typ[nd] = TYPES.get(t.value, str(t.value))
print("nodes", n.value, "edges", ne.value, collections.Counter(typ.values()))
succ, pred = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(fr, to):
    succ[a].append(b); pred[b].append(a)
print("forks (nodes with > 1 successor):", sum(len(v) > 1 for v in succ.values()), " joins (> 1 predecessor):", sum(len(v) > 1 for v in pred.values()))
print("roots:", sum(1 for nd in nodes if not pred[nd]), " leaves:", sum(1 for nd in nodes if not succ[nd]))
for nd in nodes:
    if typ[nd] == "MEMSET":
        print("MEMSET node: predecessors", [typ[p] for p in pred[nd]], "successors", [typ[s_] for s_ in succ[nd]])
