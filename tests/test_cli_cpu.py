"""CPU: the `kmerlr` command line up to the point where it needs a GPU -- the reference's option tables (getopt spelling,
unknown options end the run: kmerLr.go:85-90, kmerLr_learn.go:116-149, scoresLr_learn.go:109-136), argument checks, and
the rank launcher's failure handling (no device: every rank fails, the parent returns their status instead of hanging).
The product has no CPU fallback, so nothing past klr_ctx_create can run here."""
import os
import subprocess

import pytest

from tests.conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "kmerlr_amd", "_lib", "kmerlr")
SFG, SBG = os.path.join(GOLDEN, "scoresLr_test_fg.table"), os.path.join(GOLDEN, "scoresLr_test_bg.table")


@pytest.fixture(scope="module", autouse=True)
def built():
    from kmerlr_amd import build as B
    B.build()
    assert os.path.exists(CLI)


def run(args, timeout=120):
    return subprocess.run([CLI] + args, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("args, message", [
    (["learn", "--no-such-option", "4", "4", "a", "b", "c"], "unknown option `--no-such-option'"),
    (["--frobnicate", "learn"], "unknown option `--frobnicate'"),
    (["learn", "-x", "4", "4", "a", "b", "c"], "unknown option `-x'"),
    (["--type=scoresLr", "learn", "--binarize", "a", "b", "c"], "unknown option `--binarize' for --type=scoresLr"),
    (["learn", "--header", "4", "4", "a", "b", "c"], "unknown option `--header' for --type=kmerLr"),
    (["learn", "--lambda-auto"], "option `--lambda-auto' needs a value"),
    (["learn", "--balance=yes", "4", "4", "a", "b", "c"], "option `--balance' takes no value"),
    (["--type=other", "learn"], "invalid classifier type"),
    (["frobnicate"], "unknown command `frobnicate'"),
    (["--ranks=2", "loss", "m.json", "a", "b"], "--ranks applies to `learn'"),
    (["--ranks=0", "learn", "4", "4", "a", "b", "c"], "invalid --ranks"),
    (["--ranks=2", "--transport=carrier-pigeon", "learn", "4", "4", "a", "b", "c"], "invalid --transport"),
    (["--ranks=2", "--devices=0", "learn", "4", "4", "a", "b", "c"], "--devices needs one device per rank"),
    (["--threads=0", "learn"], "invalid number of threads"),
])
def test_bad_command_lines_end_before_the_gpu(args, message):
    r = run(args)
    assert r.returncode != 0 and message in r.stderr, (r.returncode, r.stderr)


def test_version_and_usage():
    r = run(["--version"])
    assert r.returncode == 0 and r.stdout.startswith("kmerlr-hip") and "kernels:" in r.stdout
    r = run(["--help"])
    assert r.returncode == 0 and "usage: kmerlr" in r.stdout and "--ranks=N" in r.stdout
    r = run([])
    assert r.returncode == 1 and "usage: kmerlr" in r.stdout


def no_gpu_here():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not no_gpu_here(), reason="needs a box WITHOUT a GPU: the point is the failure path")
def test_no_device_fails_loudly_also_through_the_rank_launcher(tmp_path):
    """There is no CPU fallback: one rank or several, the command ends with the library's message -- the launcher's parent
    collects its ranks' status (both transports) instead of waiting for ranks that never joined."""
    r = run(["--type=scoresLr", "learn", "--lambda-auto=2", SFG, SBG, str(tmp_path / "x")])
    assert r.returncode != 0 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr
    for transport in ("shm", "rccl"):
        r = run(["--type=scoresLr", "--ranks=2", "--transport=" + transport, "learn", "--lambda-auto=2", SFG, SBG, str(tmp_path / "y")], timeout=180)
        assert r.returncode != 0 and r.stderr.count("no HIP device") == 2 and "[rank 0]" in r.stderr and "[rank 1]" in r.stderr, r.stderr
    assert not os.path.exists(tmp_path / "x.json") and not os.path.exists(tmp_path / "y.json")


def test_command_help_lists_every_option_of_the_tables():
    """`kmerlr <command> --help` needs no device and names every option the command accepts."""
    r = run(["learn", "--help"])
    assert r.returncode == 0 and r.stdout.startswith("usage: kmerlr learn")
    for opt in ("lambda-auto", "lambda", "max-features", "balance", "co-occurrence", "co-preselection", "ensemble-size",
                "ensemble-summary", "max-epochs", "max-iterations", "max-samples", "epsilon", "epsilon-lambda", "epsilon-loss",
                "save-trace", "save-path", "eval-loss", "data-transform", "k-fold-cv", "validation-size", "scale-step-size",
                "penalty-free", "adaptive-step-size", "threads-cv", "threads-saga", "threads-lr", "alphabet", "binarize",
                "complement", "max-ambiguous", "reverse", "revcomp"):
        assert "--" + opt in r.stdout, opt
    r = run(["--type=scoresLr", "learn", "-h"])
    assert r.returncode == 0 and "--header" in r.stdout and "--binarize" not in r.stdout
    for cmd in ("loss", "predict"):
        r = run([cmd, "--help"])
        assert r.returncode == 0 and r.stdout.startswith("usage: kmerlr " + cmd)
