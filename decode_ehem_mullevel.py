#!/usr/bin/env python3
"""Drop-in for the reference's decode_ehem_mullevel.py (same flags): see scp_amd/cli.py decode_main."""
from scp_amd.cli import decode_main

if __name__ == "__main__":
    decode_main(mullevel=True)
