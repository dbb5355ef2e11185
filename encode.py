#!/usr/bin/env python3
"""Drop-in for the reference's encode.py (same flags): see scp_amd/cli.py."""
from scp_amd.cli import main

if __name__ == "__main__":
    main(mullevel=False)
