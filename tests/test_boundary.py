"""SURVEY 8(b) boundary clauses: fork safety of the zero-edit path and the explicit CPU-tensor decision.

The reference forks its DataLoader workers (second/pytorch/train.py:262-277) and the workers call
spconv.utils.VoxelGeneratorV2.generate (second/data/preprocess.py:301-316).  Here that call needs a HIP context, which does not
survive fork(): `import spconv` switches DataLoader workers to `spawn`, and a child that was nevertheless forked from a
GPU-initialised parent gets a SecondHipError with instructions from every entry point -- not a hang.
"""
import os
import pickle
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG


def _run(code, timeout=120):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, ROOT, os.environ.get("PYTHONPATH", "")]))
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, timeout=timeout, env=env)


def test_import_spconv_alone_switches_loader_workers_to_spawn():
    """No compat.install() call: the import the unmodified reference performs (middle.py:4, voxel_builder.py:3) is enough."""
    r = _run("""
        import torch.utils.data as tud
        assert not getattr(tud.DataLoader, "_second_amd_spawn", False)
        import spconv
        assert tud.DataLoader._second_amd_spawn
        class D(tud.Dataset):
            def __len__(self): return 4
            def __getitem__(self, i): return i
        class Theirs(D):
            pass
        Theirs.__module__ = "second.pytorch.builder.input_reader_builder"            # where the reference's DatasetWrapper lives
        def merge_second_batch(b): return b
        merge_second_batch.__module__ = "second.data.preprocess"
        dl = tud.DataLoader(Theirs(), num_workers=2)            # what train.py:262 does
        assert dl.multiprocessing_context.get_start_method() == "spawn", dl.multiprocessing_context
        dl = tud.DataLoader(D(), num_workers=2, collate_fn=merge_second_batch)
        assert dl.multiprocessing_context.get_start_method() == "spawn"
        dl = tud.DataLoader(Theirs(), num_workers=2, multiprocessing_context="fork")      # an explicit choice is left alone
        assert dl.multiprocessing_context.get_start_method() == "fork"
        assert tud.DataLoader(Theirs(), num_workers=0).multiprocessing_context is None
        # a loader of the host program that has nothing to do with the reference keeps its start method
        assert tud.DataLoader(D(), num_workers=2).multiprocessing_context is None
        from second_amd import compat
        compat.install()                                        # the explicit call switches every loader with workers
        assert tud.DataLoader(D(), num_workers=2).multiprocessing_context.get_start_method() == "spawn"
        print("ok")
    """)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_forked_child_of_a_gpu_parent_gets_the_message_not_a_hang():
    """CPU container: the parent's "GPU already initialised" state is simulated (torch.cuda.is_initialized patched), the fork is
    real.  Every entry point of the package must raise SecondHipError with the instructions inside the child."""
    r = _run("""
        import os, sys, pickle
        import numpy as np, torch
        import spconv
        from second_amd import runtime as rt, ops
        torch.cuda.is_initialized = lambda: True            # the parent has used the GPU
        rd, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            out = {}
            gen = spconv.utils.VoxelGeneratorV2([0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1], 5, 20000)
            for name, fn in (("generate", lambda: gen.generate(np.zeros((10, 4), np.float32), 100)),
                             ("lib", rt.lib),
                             ("rulebook", lambda: ops.rulebook_subm(torch.zeros((2, 4), dtype=torch.int32), 1, (4, 4, 4))),
                             ("nms", lambda: spconv.utils.non_max_suppression_cpu(np.zeros((2, 5), np.float32), np.arange(2, dtype=np.int32), 0.5, 0.0))):
                try:
                    fn()
                    out[name] = "no error"
                except Exception as e:
                    out[name] = (type(e).__name__, str(e))
            os.write(wr, pickle.dumps(out))
            os._exit(0)
        os.close(wr)
        data = b""
        while True:
            chunk = os.read(rd, 65536)
            if not chunk:
                break
            data += chunk
        os.waitpid(pid, 0)
        out = pickle.loads(data)
        for name, v in out.items():
            assert v[0] == "SecondHipError" and "fork" in v[1] and "spawn" in v[1], (name, v)
        # the parent itself is not poisoned
        assert not rt._forked_after_gpu_init
        print("ok", sorted(out))
    """)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_forked_child_of_a_cpu_only_parent_is_not_marked():
    r = _run("""
        import os
        import spconv
        from second_amd import runtime as rt
        pid = os.fork()
        if pid == 0:
            os._exit(7 if rt._forked_after_gpu_init else 0)
        assert os.waitpid(pid, 0)[1] == 0
        print("ok")
    """)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_cpu_tensor_refusal_names_the_decision():
    """CPU tensors are refused (documented in INTEGRATION.md section 1, "CPU tensors"): the reference's CPU device branch
    (train.py:146) cannot run its sparse middle here; the message says so instead of an attribute / pointer error."""
    import spconv
    from second_amd.runtime import SecondHipError
    x = spconv.SparseConvTensor(torch.zeros(3, 4), torch.zeros((3, 4), dtype=torch.int32), [8, 8, 8], 1)
    conv = spconv.SubMConv3d(4, 8, 3, bias=False)
    with pytest.raises(SecondHipError, match="GPU only"):
        conv(x)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "CPU tensors" in text and "train.py:146" in text


@pytest.mark.gpu
def test_fork_after_hip_init_on_hardware_raises_in_the_child():
    """The real thing: this process has a live HIP context; a fork()ed child calling the voxeliser must raise, within seconds."""
    r = _run("""
        import os, pickle, signal
        import numpy as np, torch
        import spconv
        gen = spconv.utils.VoxelGeneratorV2([0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1], 5, 20000)
        pts = np.random.default_rng(0).uniform([0, -40, -3, 0], [70, 40, 1, 1], (1000, 4)).astype(np.float32)
        n_parent = gen.generate(pts, 20000)["voxel_num"]            # initialises HIP in the parent
        assert n_parent > 0
        rd, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            signal.alarm(30)
            try:
                gen.generate(pts, 20000)
                msg = ("no error", "")
            except Exception as e:
                msg = (type(e).__name__, str(e))
            os.write(wr, pickle.dumps(msg))
            os._exit(0)
        os.close(wr)
        data = os.read(rd, 65536)
        os.waitpid(pid, 0)
        name, text = pickle.loads(data)
        assert name == "SecondHipError" and "fork" in text, (name, text)
        assert gen.generate(pts, 20000)["voxel_num"] == n_parent    # the parent keeps working
        print("ok")
    """, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_capture_guard_keeps_the_collector_off_and_restores_it():
    """runtime.capture_guard: garbage is collected BEFORE the guarded region (a hipGraph destroyed while a stream captures aborts the process),
    the collector is off inside it -- also when the region raises -- and back in its previous state afterwards."""
    import gc
    import weakref
    from second_amd import runtime as rt

    class Node:
        pass
    a, b = Node(), Node()
    a.other, b.other = b, a                  # a reference cycle: only the collector frees it
    probe = weakref.ref(a)
    del a, b
    assert gc.isenabled()
    with rt.capture_guard():
        assert probe() is None               # collected on entry
        assert not gc.isenabled()
    assert gc.isenabled()
    try:
        with rt.capture_guard():
            raise ValueError("inside")
    except ValueError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with rt.capture_guard():
            assert not gc.isenabled()
        assert not gc.isenabled()            # it was off before: stays off
    finally:
        gc.enable()
