import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pb_oracle
    pb_oracle.build()
    return pb_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


PARITY_LINES = []          # "[every-read parity] ..." lines of the full-size GPU tests, shown in the run's summary


def note_parity(line):
    PARITY_LINES.append(line)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "every_read_parity.txt"), "a") as f:
            f.write(line + "\n")


def pytest_terminal_summary(terminalreporter):
    for line in PARITY_LINES:
        terminalreporter.write_line(line)
