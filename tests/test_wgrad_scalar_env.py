"""CPU: the SF_WGRAD_SCALAR=1 fragment path (read once per process) through the host simulator."""
import os
import subprocess
import sys


def test_wgrad_scalar_path(hostsim_path):
    code = ("import torch; from tests import kernel_checks as kc; d=torch.device('cpu');"
            "kc.check_conv_wgrad(d,(1,16,1,9,9),24,(1,3,3),(1,2,2),(0,1,1)); print('ok')")
    env = dict(os.environ, SF_WGRAD_SCALAR="1", SFAMD_LIBRARY=hostsim_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_depthwise_version1_kernels(hostsim_path):
    """SF_DW_FWD_V2=0 / SF_DW_DGRAD_V2=0 / SF_DW_WGRAD_V2=0 select the first-generation W-blocked depthwise stencils (kept
    for A/B runs against the version-2 kernels that are the default)."""
    code = ("import torch; from tests import token_checks as tc; d=torch.device('cpu');"
            "tc.check_dwconv(d,2,2,16,(2,6,6),(3,3,3),(1,2,2),cls=1);"
            "tc.check_dwconv(d,1,1,16,(3,5,8),(3,3,3),(1,1,1),cls=0);"
            "tc.check_dwconv(d,1,1,120,(2,4,8),(3,3,3),(1,1,1),cls=0); print('ok')")
    env = dict(os.environ, SF_DW_FWD_V2="0", SF_DW_DGRAD_V2="0", SF_DW_WGRAD_V2="0", SFAMD_LIBRARY=hostsim_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
