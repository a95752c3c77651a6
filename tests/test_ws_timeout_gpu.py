"""A bounded wait of the wave-specialised SepConv kernels that gives up must not pass for a result (csrc/sepconv_ws.hip, SURVEY 5 "race
detection"): the kernels never hang, they finish with wrong numbers and a count -- and the product raises BEFORE the meta-iteration's
outer optimizer step (theta and the optimizer state stay untouched), clears the polled word, and runs the next iteration normally.
The process total is per process, so the provoked failure runs in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, sys
sys.path.insert(0, %r)
import torch
from meta_interpolation_amd import _hip, synthetic
from tests.helpers import build_system, golden, parse_case_args
fx = golden("system_sepconv_lslr_sgd_2step")
system = build_system("sepconv", parse_case_args(fx))
frames = synthetic.septuplet_batch(int(fx['B']), int(fx['H']), int(fx['W']), model="sepconv")
lib = _hip.lib()
losses, _, _ = system.run_train_iter(data_batch=frames, epoch=0)
float(losses['loss'])                                    # a healthy iteration reads fine
assert lib.savfi_sepconv_ws_errors_peek() == 0 and lib.savfi_sepconv_ws_errors() == 0
prev = ctypes.c_int(0)
theta = {k: v.detach().clone() for k, v in system.state_dict().items()}
opt_state = repr(system.optimizer.state_dict()['state'])
assert lib.savfi_sepconv_ws_debug_spin_limit(-1, ctypes.byref(prev)) == 0 and prev.value == 1 << 19
try:
    losses, _, _ = system.run_train_iter(data_batch=frames, epoch=0)
except _hip.SavfiHipError as exc:
    # fail closed: the iteration raised BEFORE its outer optimizer step -- theta and the optimizer state are what they were
    assert "gave up" in str(exc) and "NOT applied" in str(exc), exc
    torch.cuda.synchronize()
    after = system.state_dict()
    assert all(torch.equal(theta[k], after[k]) for k in theta), "theta moved in an iteration whose waits gave up"
    assert repr(system.optimizer.state_dict()['state']) == opt_state
    assert lib.savfi_sepconv_ws_errors() > 0            # the process total keeps the count ...
    assert lib.savfi_sepconv_ws_errors_peek() == 0      # ... the word the product polls was cleared when the error was reported
    # handled: with the waits back to normal the next iteration runs and moves theta
    assert lib.savfi_sepconv_ws_debug_spin_limit(prev.value, None) == 0
    system.optimizer.zero_grad()
    losses, _, _ = system.run_train_iter(data_batch=frames, epoch=0)
    float(losses['loss'])
    torch.cuda.synchronize()
    assert any(not torch.equal(theta[k], v) for k, v in system.state_dict().items())
    print("RAISED")
    sys.exit(0)
print("no exception: an iteration whose waits gave up stepped the outer optimizer")
sys.exit(2)
''' % ROOT


def test_a_wait_that_gives_up_raises_before_the_outer_step_and_leaves_theta_untouched():
    out = subprocess.run([sys.executable, '-c', CHILD], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == 'RAISED', out.stdout[-2000:] + out.stderr[-3000:]
