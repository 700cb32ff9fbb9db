"""N > 1 path on CPU: world_size-2 torch.distributed (gloo) runs of the sharded Simulation.

Each rank owns an even-aligned range of GLOBAL chain ids; sweeps need no collective; callbacks and
the estimator fold are one all-reduce(sum).  Because the Philox counter is keyed by the global chain
id, every callback row and the learned sigma must equal the single-process run (shard invariance).
The engine is the oracle double (no GPU here); the sharding / all-reduce code under test is the
product's (montecarlo_amd/sharding.py, metropolis.py, policy_guided.py).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch.distributed as dist
import montecarlo_amd as ma
import oracle_lib as O

world = int(os.environ.get("WORLD_SIZE", "1"))
use_store = os.environ.get("AMC_TEST_GROUP") == "store"       # a key-value store only (the package's plain-socket one): no process group at all
if world > 1 and use_store:
    from montecarlo_amd import sharding
    rank = sharding.init_store_group().rank
elif world > 1:
    dist.init_process_group("gloo")
    rank = dist.get_rank()
else:
    rank = 0
out = sys.argv[1]
M, steps = 37, 60
chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.6),
        ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 0.4))
learn = ma.VPG(0.01) if os.environ.get("AMC_TEST_LEARN", "1") == "1" else ma.Static()
if os.environ.get("AMC_TEST_POLICY") == "drift2":
    # a policy with TWO parameters (delta = theta0 + theta1 z): GradientData rows of 2 + 2P + P^2 records cross the shards, the
    # natural-gradient step inverts the 2 x 2 metric on every rank from the same merged sums
    pol = ma.ScriptPolicy("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
                          ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"], n_params=2)
    pool = (ma.Move(ma.Displacement(), pol, [0.0, 0.2], 0.6), ma.Move(ma.Displacement(), pol, [0.05, 0.1], 0.4))
    learn = ma.NPG(0.002, 1e-6)
al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42, engine_factory=O.OracleEngine),
      dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), learn), q_batch_size=2),
      dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,), scheduler=ma.build_schedule(steps, 10, 2)),
      dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance, ma.callback_moments),
           scheduler=ma.build_schedule(steps, 10, 5)))
sim = ma.Simulation(chains, al, steps, path=os.path.join(out, "data"))
ma.run(sim)
cb = sim.algorithms[-1]
res = dict(rank=rank, shard=list(sim.algorithms[0].shard), x=[v.hex() for v in chains.x],
           energy=[(t, float(v)) for t, v in cb.rows[0]], acceptance=[(t, [float(a) for a in v]) for t, v in cb.rows[1]],
           sigma=[float(m.sigma) for m in pool], parameters=[[float(v).hex() for v in m.parameters] for m in pool], accepted=[m.accepted_calls for m in pool], total=[m.total_calls for m in pool])
json.dump(res, open(os.path.join(out, f"rank{{rank}}.json"), "w"))
if world > 1 and use_store:
    sharding.barrier()
elif world > 1:
    dist.destroy_process_group()
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(tmp_path, world, learn=True, group="gloo", policy=""):
    out = tmp_path / f"w{world}{'L' if learn else 'S'}{group}{policy}"
    out.mkdir()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1", AMC_TEST_LEARN="1" if learn else "0", AMC_TEST_GROUP=group, AMC_TEST_POLICY=policy)
    if world == 1:
        cmd = [sys.executable, str(script), str(out)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [json.load(open(out / f"rank{k}.json")) for k in range(world)]


@pytest.mark.slow
def test_two_ranks_reproduce_one_rank(tmp_path):
    one = run_world(tmp_path, 1)[0]
    two = run_world(tmp_path, 2)
    assert two[0]["shard"] == [0, 20] and two[1]["shard"] == [20, 37]          # even boundary
    # per-chain states: sigma is learned from cross-shard sums, and those are reproducible sums (DESIGN.md section 3.8:
    # integer records merged exactly, rounded once) -- nothing depends on how the chains are split
    assert two[0]["x"] + two[1]["x"] == one["x"]
    for r in two:
        # every rank sees the same global callback values
        assert [t for t, _ in r["energy"]] == [t for t, _ in one["energy"]]
        assert r["energy"] == one["energy"]
        assert np.array_equal(np.array([v for _, v in r["acceptance"]]), np.array([v for _, v in one["acceptance"]]), equal_nan=True)
        # sigma stays replicated without a broadcast: same merged GradientData -> same learning_step!
        assert r["sigma"] == one["sigma"]
        assert r["sigma"][0] == 0.2 and r["sigma"][1] != 0.1
        assert r["accepted"] == one["accepted"] and r["total"] == one["total"]
    assert two[0]["energy"] == two[1]["energy"]


@pytest.mark.slow
def test_two_ranks_reproduce_one_rank_with_a_two_parameter_policy(tmp_path):
    """The same with a policy of two parameters and NPG: rows of ten records per learnable move cross the shards, both ranks
    invert the same 2 x 2 metric -- parameter vectors, callback rows and chains equal to the one-rank run, bit for bit."""
    one = run_world(tmp_path, 1, policy="drift2")[0]
    two = run_world(tmp_path, 2, group="store", policy="drift2")
    assert two[0]["x"] + two[1]["x"] == one["x"]
    for r in two:
        assert r["parameters"] == one["parameters"] and r["energy"] == one["energy"]
        assert r["parameters"][0] == [float(0.0).hex(), float(0.2).hex()] and r["parameters"][1] != [float(0.05).hex(), float(0.1).hex()]


def test_socket_store_steps_aside_for_a_stranger_on_its_port():
    """MASTER_PORT + 1 owned by something else: rank 0 binds the next candidate port, the clients pass the stranger by (it does
    not greet them as the store does) and find the store."""
    import threading
    from montecarlo_amd.sharding import SocketStore
    stranger = socket.socket()
    stranger.bind(("127.0.0.1", 0))
    stranger.listen(8)
    port = stranger.getsockname()[1]
    threading.Thread(target=lambda: [stranger.accept()[0].sendall(b"HTTP/1.0 400 go away\r\n\r\n") for _ in range(4)], daemon=True).start()
    master = SocketStore("127.0.0.1", port, is_master=True, timeout_s=20.0)
    client = SocketStore("127.0.0.1", port, is_master=False, timeout_s=20.0)
    master.set("k", b"v")
    assert client.get("k") == b"v" and client.add("n", 2) == 2 and master.add("n", 3) == 5
    assert master._srv.getsockname()[1] == port + SocketStore.PORT_OFFSETS[1]
    master._srv.close()
    stranger.close()


def test_socket_store_serves_its_own_launch_only(tmp_path):
    """Both ends prove the launch token before anything is served: a client of another launch does not take the server for its
    store (it keeps looking and times out), and one that skips the handshake is dropped unserved; the rightful client works
    before and after."""
    import struct
    from montecarlo_amd import sharding as S
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    master = S.SocketStore("127.0.0.1", port, is_master=True, timeout_s=20.0, token=b"launch-A")
    master.set("k", b"v")
    with pytest.raises(TimeoutError):
        S.SocketStore("127.0.0.1", port, is_master=False, timeout_s=0.5, token=b"launch-B")
    # a peer that reads the greeting and then talks without answering it: a would-be pickle bomb in the old framing
    marker = tmp_path / "executed"
    import pickle

    class Bomb:
        def __reduce__(self):
            return (open, (str(marker), "w"))
    payload = pickle.dumps(("set", "x", Bomb()))
    raw = socket.create_connection(("127.0.0.1", port), timeout=5.0)
    raw.recv(4096)
    raw.sendall(struct.pack("<Q", len(payload)) + payload)
    raw.settimeout(5.0)
    try:
        assert raw.recv(16) == b""                      # dropped: nothing parsed, nothing answered
    except ConnectionResetError:                        # (closed with our bytes still unread)
        pass
    raw.close()
    assert not marker.exists()
    friend = S.SocketStore("127.0.0.1", port, is_master=False, timeout_s=20.0, token=b"launch-A")
    assert friend.get("k") == b"v" and friend.num_keys() == 1
    # after the handshake: frames are plain fields; garbage and oversized frames end the connection, the store lives on
    friend._sock.sendall(struct.pack("<I", S.SocketStore.MAX_FRAME + 1))
    with pytest.raises((ConnectionError, OSError)):
        friend.get("k")
    again = S.SocketStore("127.0.0.1", port, is_master=False, timeout_s=20.0, token=b"launch-A")
    assert again.get("k") == b"v"
    with pytest.raises(TypeError):
        again.set("obj", {"not": "bytes"})
    master._srv.close()


def test_socket_store_handshake_is_a_challenge_and_a_response(monkeypatch):
    """Nothing on the wire is a fixed function of the token: the server greets with a fresh nonce, the client answers with its own
    nonce and an HMAC over both, the server proves itself the same way.  A stranger who connects learns a nonce and nothing to
    brute-force offline; a replay of an honest client's answer on another connection fails (another nonce); an impostor server
    that cannot prove the token is passed by.  A store other hosts can reach refuses to start without AMC_STORE_TOKEN, and the
    timeout of a rank that cannot find its store says what the token is made of."""
    import hashlib
    import hmac
    import threading
    from montecarlo_amd import sharding as S
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    master = S.SocketStore("127.0.0.1", port, is_master=True, timeout_s=20.0, token=b"launch-A")
    head = f"{S.SocketStore.MAGIC} {port} ".encode()
    greetings = []
    for _ in range(2):
        raw = socket.create_connection(("127.0.0.1", port), timeout=5.0)
        raw.settimeout(5.0)
        g = raw.recv(4096)
        assert g.startswith(head) and g.endswith(b"\n") and len(g) == len(head) + 33
        greetings.append((raw, g[len(head):-1]))
    assert greetings[0][1] != greetings[1][1]                                   # a nonce per connection
    token_digests = [hashlib.sha256(w + b"\0launch-A").hexdigest().encode() for w in (b"server", b"client")]
    assert all(d not in g for d in token_digests for _, g in greetings)           # (what round 5's greeting was)
    # an honest answer for connection 0 ...
    sn0, cn = greetings[0][1], b"0123456789abcdef0123456789abcdef"
    mac0 = hmac.new(b"launch-A", b"client\0" + sn0 + b"\0" + cn, hashlib.sha256).hexdigest().encode()
    # ... replayed on connection 1 is refused: dropped without a proof
    greetings[1][0].sendall(cn + mac0 + b"\n")
    try:
        assert greetings[1][0].recv(128) == b""
    except ConnectionResetError:
        pass
    # on its own connection it is accepted, and the server proves that it holds the token too
    greetings[0][0].sendall(cn + mac0 + b"\n")
    proof = greetings[0][0].recv(128)
    assert proof == hmac.new(b"launch-A", b"server\0" + sn0 + b"\0" + cn, hashlib.sha256).hexdigest().encode() + b"\n"
    for raw, _ in greetings:
        raw.close()
    master._srv.close()
    # an impostor that speaks the greeting but cannot prove the token: the client does not take it for the store
    imp = socket.socket()
    imp.bind(("127.0.0.1", 0))
    imp.listen(8)
    iport = imp.getsockname()[1]

    def impostor():
        try:
            while True:
                c, _ = imp.accept()
                c.sendall(f"{S.SocketStore.MAGIC} {iport} ".encode() + b"f" * 32 + b"\n")
                c.recv(4096)
                c.sendall(b"0" * 64 + b"\n")
                c.close()
        except OSError:
            pass
    threading.Thread(target=impostor, daemon=True).start()
    monkeypatch.delenv("AMC_STORE_TOKEN", raising=False)
    with pytest.raises(TimeoutError, match="AMC_STORE_TOKEN") as ei:
        S.SocketStore("127.0.0.1", iport, is_master=False, timeout_s=0.6, token=b"launch-A")
    assert "TORCHELASTIC_RUN_ID" in str(ei.value) and "WORLD_SIZE" in str(ei.value)
    imp.close()
    # reachable from other hosts: only with a secret
    with pytest.raises(OSError, match="without AMC_STORE_TOKEN"):
        S.SocketStore("10.255.255.1", port, is_master=True, timeout_s=0.5)
    # ranks started by hand (no launcher's run id) do not mix the parent's pid into the token; under the launcher they do
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert S.SocketStore.launch_token("127.0.0.1", 29500) == b"//2/29500"
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    assert S.SocketStore.launch_token("127.0.0.1", 29500) == f"none//2/29500/{os.getppid()}".encode()


def test_store_values_are_plain_data():
    """What the ranks exchange through the store is a closed set of plain types in a tagged binary form -- round trip, and no
    way in for objects."""
    from montecarlo_amd.sharding import _dumps, _loads
    vals = [None, True, False, 7, -2**62, 0.1, float("inf"), "reason: \u00e9", bytes(range(128)), [1, 2.5, "x"], (True, None),
            {"rank": 1, "v": [1, 1, 1], "nested": {"a": (1, b"z")}}, np.arange(12, dtype=np.float64).reshape(3, 4),
            np.array([], dtype=np.int32), np.float64(2.5), np.int64(-3), np.bool_(True)]
    for v in vals:
        got = _loads(_dumps(v))
        if isinstance(v, np.ndarray):
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
        else:
            assert got == v and type(got) is (type(v) if not isinstance(v, np.generic) else type(v.item()))
    nan = _loads(_dumps(float("nan")))
    assert nan != nan
    for bad in (object(), {1, 2}, np.array([object()]), lambda: 0):
        with pytest.raises(TypeError):
            _dumps(bad)
    for junk in (b"", b"?", b"i\x00", b"a\x03\x01|O8" + b"\x00" * 8, _dumps(1) + b"x"):
        with pytest.raises((ValueError, TypeError, Exception)):
            _loads(junk)


@pytest.mark.slow
def test_two_ranks_bit_exact_without_learning(tmp_path):
    """With every optimiser Static nothing the chains see depends on a cross-shard sum: the concatenated shards equal
    the unsharded run bit for bit (Philox keyed by the GLOBAL chain id)."""
    one = run_world(tmp_path, 1, learn=False)[0]
    two = run_world(tmp_path, 2, learn=False)
    assert two[0]["x"] + two[1]["x"] == one["x"]
    for r in two:
        assert r["sigma"] == [0.2, 0.1] and r["accepted"] == one["accepted"] and r["total"] == one["total"]


@pytest.mark.slow
def test_two_ranks_over_the_launchers_store_only(tmp_path):
    """The same sharded run with NO process group: ranks meet through torch.distributed.run's TCP store
    (sharding.init_store_group), sums of engines without a communicator are added on the host in rank order.  This is
    the plumbing bench.py uses for N > 1 (there the sums go through the engines' own RCCL communicator)."""
    one = run_world(tmp_path, 1, learn=False)[0]
    two = run_world(tmp_path, 2, learn=False, group="store")
    assert two[0]["shard"] == [0, 20] and two[1]["shard"] == [20, 37]
    assert two[0]["x"] + two[1]["x"] == one["x"]
    for r in two:
        assert r["energy"] == one["energy"]
        assert r["accepted"] == one["accepted"] and r["total"] == one["total"]
    assert two[0]["energy"] == two[1]["energy"]
    learned = run_world(tmp_path, 2, learn=True, group="store")
    ref = run_world(tmp_path, 2, learn=True)
    for a, b in zip(learned, ref):
        assert a["sigma"] == b["sigma"]       # gloo's all-reduce or the store's gather: records are merged exactly either way


@pytest.mark.parametrize("store", ["socket", "torch"])
def test_store_group_primitives_two_processes(tmp_path, store, monkeypatch):
    """barrier / allgather / broadcast / allreduce_sum / allreduce_xsum of StoreGroup -- over the package's own plain-socket store
    (the default: the worker never imports torch) and over the launcher's TCPStore (AMC_STORE=torch), under the launcher and
    started by hand."""
    monkeypatch.setenv("AMC_STORE", store)
    script = tmp_path / "sg.py"
    script.write_text(f"""
import os, sys, json
sys.path.insert(0, {ROOT!r})
import numpy as np
from montecarlo_amd import sharding
g = sharding.init_store_group()
assert g.kind == {store!r} and (("torch" in sys.modules) == (g.kind == "torch"))
assert sharding.world() == (g.rank, 2)
# records of reproducible sums: every rank ends with the merged records of all (a gather + exact merges)
from montecarlo_amd import _capi
rec = _capi.xsum_plain([1.0 + g.rank, 10.0])
assert list(_capi.xsum_round(sharding.allreduce_xsum(rec))) == [3.0, 20.0]
g.barrier()
got = g.allgather(dict(rank=g.rank, v=[g.rank] * 3))
assert [d["rank"] for d in got] == [0, 1]
uid = g.broadcast(bytes(range(128)) if g.rank == 0 else None)
assert uid == bytes(range(128))
s = sharding.allreduce_sum(np.array([1.0 + g.rank, 0.5, -g.rank]))
assert list(s) == [3.0, 1.0, -1.0]
# a loop whose exit condition is local (bench.py's spin-up clock): the ranks leave after the same number of rounds
rounds = 0
while True:
    rounds += 1
    if sharding.all_ranks(rounds >= 2 + 3 * g.rank):
        break
assert g.allgather(rounds) == [5, 5]
sharding.barrier()
json.dump(dict(ok=True), open(os.path.join({str(tmp_path)!r}, f"sg{{g.rank}}.json"), "w"))
""")
    # (a) under the driver's launcher: the agent hosts the store, the workers are its clients
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert all(json.load(open(tmp_path / f"sg{k}.json"))["ok"] for k in range(2))
    for k in range(2):
        os.remove(tmp_path / f"sg{k}.json")
    # (b) started by hand: rank 0 hosts the store
    port = free_port()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                                                     MASTER_PORT=str(port)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
    assert all(json.load(open(tmp_path / f"sg{r}.json"))["ok"] for r in range(2))


@pytest.mark.parametrize("store", ["socket", "torch"])
def test_connect_engine_is_all_or_none_and_store_keys_are_retired(tmp_path, store, monkeypatch):
    """sharding.connect_engine is a collective: whatever fails on whichever rank (no librccl on rank 0, ncclCommInitRank
    failing on one rank only), every rank consumes the same store rounds, gets the same answer and is left a single shard,
    and the host-side sums that follow stay in step.  Also: the launcher's store does not grow with the number of sums."""
    monkeypatch.setenv("AMC_STORE", store)
    script = tmp_path / "ce.py"
    script.write_text(f"""
import os, sys, json
sys.path.insert(0, {ROOT!r})
import numpy as np
from montecarlo_amd import sharding
g = sharding.init_store_group()

class Double:
    def __init__(self, uid_fails=False, init_fails_on=None):
        self.uid_fails, self.init_fails_on, self.destroyed, self.inited = uid_fails, init_fails_on, 0, 0
    def comm_unique_id(self):
        if self.uid_fails:
            raise RuntimeError("cannot dlopen librccl")
        return bytes(128)
    def comm_init(self, rank, size, uid):
        assert uid == bytes(128) and size == 2
        if self.init_fails_on == rank:
            raise RuntimeError("ncclCommInitRank failed")
        self.inited += 1
    def comm_destroy(self):
        self.destroyed += 1
    def allreduce_sum(self, v):
        raise AssertionError("a disconnected engine must not be asked")

# (a) rank 0 cannot even make the id: nobody waits for a key that never comes
e = Double(uid_fails=True)
assert sharding.connect_engine(e) is False and e.comm_connected is False and e.inited == 0
assert list(sharding.allreduce_sum(np.array([1.0 + g.rank]), e)) == [3.0]          # same round on both ranks
# (b) comm_init fails on rank 1 only: rank 0 drops the communicator it got
e = Double(init_fails_on=1)
assert sharding.connect_engine(e) is False and e.comm_connected is False
assert (e.inited, e.destroyed) == ((1, 1) if g.rank == 0 else (0, 0))
assert list(sharding.allreduce_sum(np.array([2.0 * g.rank]), e)) == [2.0]
# (c) everything works: connected on both
e = Double()
assert sharding.connect_engine(e) is True and e.comm_connected is True and e.destroyed == 0
# (d) an engine that cannot hold a communicator at all (CPU test doubles)
assert sharding.connect_engine(object.__new__(type("Bare", (), {{}}))) is False
# the store holds a bounded number of keys however many sums have passed
g.barrier()
before = g.store.num_keys()
for i in range(200):
    s = g.allreduce_sum(np.array([float(i), 1.0]))
    assert list(s) == [2.0 * i, 2.0]
    if i % 50 == 0:
        assert g.broadcast(i if g.rank == 0 else None) == i
g.barrier()
after = g.store.num_keys()
assert after <= before + 16, (before, after)
json.dump(dict(ok=True, keys=[before, after]), open(os.path.join({str(tmp_path)!r}, f"ce{{g.rank}}.json"), "w"))
g.barrier()
""")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert all(json.load(open(tmp_path / f"ce{k}.json"))["ok"] for k in range(2))
