#!/usr/bin/env python3
"""moira.py -- same command line as the reference script, MI355X filter underneath.
See moira_amd/cli.py."""
import sys

from moira_amd.cli import cli

if __name__ == "__main__":
    sys.exit(cli())
