"""bench.py as the driver invokes it (CPU-only checks): `--gpus N` without WORLD_SIZE becomes the launcher of N ranks - a plain child process
running `python -m torch.distributed.run ... bench.py <same flags>` on 127.0.0.1 - before anything imports torch.cuda, and relays the exit code."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_n_self_launches_ranks_and_relays_exit_code(monkeypatch):
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return Done()
    monkeypatch.setattr(bench.subprocess, 'run', fake_run)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                    # the ranks' exit code is ours
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and 0 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']          # the same flags reach every rank
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_rank_count_mismatch_is_refused(monkeypatch):
    import bench
    monkeypatch.setenv('WORLD_SIZE', '2'); monkeypatch.setenv('RANK', '0'); monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4'])
    with pytest.raises(SystemExit, match='WORLD_SIZE=2'):
        bench.main()


def test_host_cores_respects_override(monkeypatch):
    import bench
    monkeypatch.setenv('DSRL_CPU_BASELINE_THREADS', '3')
    assert bench.host_cores() == 3
