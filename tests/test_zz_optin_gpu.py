"""GPU (-m gpu), opt-in code paths: kept in a file that sorts last, so that with ``pytest -x`` a failure of code that is not
on the default path cannot hide the results of the default-path tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_depthwise_version2_kernels_gpu(gpu):
    """The version-2 depthwise stencils (SF_DW_*_V2=1, profiles/r1/r1_isa_dwconv_v2.md) on the real kernels: narrow (fp32 LDS
    weights) and wide (fp16) layers, stride 1 and 2, whole and ragged 4-column groups, cls rows; then X3D end to end."""
    code = ("import torch; from tests import token_checks as tc, model_checks as mc; d=torch.device('cuda:0');"
            "tc.check_dwconv(d,2,2,96,(4,14,14),(3,3,3),(1,2,2),cls=1);"
            "tc.check_dwconv(d,1,4,96,(4,7,7),(3,3,3),(1,1,1),cls=1);"
            "tc.check_dwconv(d,2,1,24,(8,12,12),(5,1,1),(1,1,1),cls=0);"
            "tc.check_dwconv(d,2,1,216,(4,14,14),(3,3,3),(1,2,2),cls=0);"
            "tc.check_dwconv(d,2,1,56,(4,28,28),(3,3,3),(1,1,1),cls=0);"
            "tc.check_dwconv(d,2,1,56,(4,56,56),(3,3,3),(1,2,2),cls=0);"
            "tc.check_dwconv(d,1,1,432,(4,8,8),(3,3,3),(1,1,1),cls=0);"
            "mc.check_engine('x3d_tiny', d, loss_scale=1.0, tol_logits=4e-3, tol_loss=1e-3, tol_gnorm=2e-3, tol_param=0.1,"
            " tol_global=1e-2); print('ok')")
    env = dict(os.environ, SF_DW_FWD_V2="1", SF_DW_DGRAD_V2="1", SF_DW_WGRAD_V2="1")
    env.pop("SFAMD_LIBRARY", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
