"""The Java side of the boundary as FILES (jni/wittgpu_jni.c, java/net/consensys/wittgenstein/core/gpu/*.java). No JDK
exists in the build image, so what can be checked here is checked: the shim is well-formed C against the JNI call shapes
(tests/c/jni_min/jni.h — a test-only declaration of the members it uses, not the JDK header) and against
include/wittgpu.h; it binds EVERY export of the two headers; and it and WittGpu.java declare the same native methods
with the same number of parameters. The ABI itself is exercised by tests/c/test_abi_full.c and the ctypes tests."""
import os
import re
import subprocess

from wittgenstein_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "jni", "wittgpu_jni.c")
JAVA = os.path.join(ROOT, "java", "net", "consensys", "wittgenstein", "core", "gpu")


def test_shim_is_well_formed_c_against_the_jni_call_shapes_and_the_abi_header():
    subprocess.run(["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "c", "jni_min"),
                    "-I" + os.path.join(ROOT, "include"), SHIM], check=True)


def test_shim_binds_every_export_of_the_two_headers():
    src = open(SHIM).read()
    missing = [s for s in _lib.ABI_SYMBOLS if not re.search(r"\b%s\b" % s, src)]  # (called, or handed to the HOST_CREATE macro)
    assert not missing, missing
    # ... and the headers declare nothing the binding's list lacks
    decl = set()
    for h in ("wittgpu.h", "wittgpu_host.h"):
        decl |= set(re.findall(r"\b(wgh?_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    decl -= {"wg_allreduce_fn"}
    assert decl <= set(_lib.ABI_SYMBOLS), sorted(decl - set(_lib.ABI_SYMBOLS))


def _java_natives():
    src = open(os.path.join(JAVA, "WittGpu.java")).read()
    out = {}
    for m in re.finditer(r"public static native\s+[\w\[\]\.]+\s+(\w+)\(([^)]*)\);", src):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def _shim_natives():
    src = open(SHIM).read()
    out = {}
    for m in re.finditer(r"WG_JNI\(\w+,\s*(\w+)\)\(JNIEnv\* env, jclass c([^)]*)\)", src):
        out[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip()])
    out.pop("NAME", None)  # (the HOST_CREATE macro's own definition)
    for m in re.finditer(r"^HOST_CREATE\((\w+),", src, re.M):  # the macro's natives: (params, nb, nl, seed, cfgInts, cfgLongs, rcclId)
        out[m.group(1)] = 7
    return out


def test_java_natives_and_shim_functions_agree():
    j, c = _java_natives(), _shim_natives()
    assert len(j) > 60
    assert set(j) == set(c), (sorted(set(j) - set(c)), sorted(set(c) - set(j)))
    bad = {k: (j[k], c[k]) for k in j if j[k] != c[k]}
    assert not bad, bad


def test_gpu_network_overrides_the_reference_surface():
    """GpuNetwork keeps the method names and argument order protocols call (C/Network.java:341-447, 505-531, 304-338)"""
    src = open(os.path.join(JAVA, "GpuNetwork.java")).read()
    for sig in ["public void send(Message<? extends TN> mc, int sendTime, TN fromNode, TN toNode)",
                "public void send(Message<? extends TN> m, int sendTime, TN fromNode, List<? extends Node> dests, int delaysBetweenMessage)",
                "public void sendArriveAt(Message<? extends TN> mc, int arriveAt, TN fromNode, TN toNode)",
                "public void registerTask(final Runnable task, int startAt, TN fromNode)",
                "public void registerPeriodicTask(final Runnable task, int startAt, int period, TN fromNode)",
                "public void registerPeriodicTask(final Runnable task, int startAt, int period, TN fromNode, Condition c)",
                "public void registerConditionalTask(final Runnable task, int startAt, int duration, TN fromNode, Condition startIf, Condition repeatIf)",
                "public boolean runMs(int ms)", "public void partition(float part)", "public void endPartition()"]:
        assert sig in src, sig


def test_gpu_network_handles_and_deferred_init_rd():
    """ADVICE.md round 5, at source level (no JDK in the image): every native call that creates an envelope runs under
    underHandle — the reference handleOf took is given back when the engine refuses the call — and deferredInit ends by putting
    the engine's rd where init() left Java's (rdToEngine() after the replay loop). The behaviour itself is tested on the Python
    mirror of the same binding: tests/test_gpu_hostmode.py::test_a_refused_send_gives_its_handle_back,
    ::test_deferred_init_leaves_rd_where_init_left_it."""
    src = open(os.path.join(JAVA, "GpuNetwork.java")).read()
    for call in ("WittGpu.send(", "WittGpu.sendArriveAt(", "WittGpu.registerTask("):
        sites = [m.start() for m in re.finditer(re.escape(call), src)]
        assert sites
        for at in sites:
            line = src[src.rfind("\n", 0, at) + 1:src.find("\n", at)]
            assert "underHandle(" in line and "handleOf(" not in line, line
    body = src[src.index("public void deferredInit(Runnable init)"):]
    body = body[:body.index("@Override")]
    assert body.rstrip().endswith("rdToEngine();\n  }".rstrip()) or body.rstrip().rstrip("}").rstrip().endswith("rdToEngine();")
    assert body.index("for (final Object[] o : kept)") < body.rindex("rdToEngine();")
