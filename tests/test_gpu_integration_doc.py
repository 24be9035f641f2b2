"""The ctypes stub that the integration guide at the repository root prints is run as it stands."""
import os
import re

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_minimal_ctypes_stub_from_the_integration_notes_runs():
    import torch

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 2. Minimal ctypes stub"):text.index("## 3.")]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    cwd = os.getcwd()
    os.chdir(ROOT)  # the stub loads the library by its repo-relative path
    try:
        scope = {}
        exec(compile(code, "INTEGRATION.md", "exec"), scope)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    obs, rew, trunc = scope["obs"], scope["rew"], scope["trunc"]
    assert obs.shape == (32, 202) and torch.isfinite(obs).all() and obs.abs().max() <= 30
    # one-hot facing direction of the row's own agent
    assert torch.equal(obs[:, 0:4].sum(1), torch.ones(32, device="cuda"))
    assert (rew >= 0).all() and not trunc.any()
