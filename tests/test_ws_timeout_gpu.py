"""A bounded wait of the wave-specialised SepConv kernels that gives up must not pass for a result (csrc/sepconv_ws.hip, SURVEY 5 "race
detection"): the kernels never hang, they finish with wrong numbers and a count -- and the product raises when it reads the iteration's
loss.  The count is per process, so the provoked failure runs in a child process."""
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
assert lib.savfi_sepconv_ws_debug_spin_limit(-1, ctypes.byref(prev)) == 0 and prev.value == 1 << 19
losses, _, _ = system.run_train_iter(data_batch=frames, epoch=0)
try:
    float(losses['loss'])
except _hip.SavfiHipError as exc:
    assert "gave up" in str(exc), exc
    assert lib.savfi_sepconv_ws_errors_peek() > 0 and lib.savfi_sepconv_ws_errors() == lib.savfi_sepconv_ws_errors_peek()
    # and it keeps refusing: the next iteration does not start
    assert lib.savfi_sepconv_ws_debug_spin_limit(prev.value, None) == 0
    try:
        system.run_train_iter(data_batch=frames, epoch=0)
    except _hip.SavfiHipError:
        print("RAISED")
        sys.exit(0)
    print("the next iteration started after a reported time-out")
    sys.exit(3)
print("no exception: the loss of an iteration whose waits gave up was handed out")
sys.exit(2)
''' % ROOT


def test_a_wait_that_gives_up_raises_when_the_loss_is_read():
    out = subprocess.run([sys.executable, '-c', CHILD], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == 'RAISED', out.stdout[-2000:] + out.stderr[-3000:]
