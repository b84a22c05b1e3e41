"""`python -m zeekstd_amd ...` == the zeekstd command line (zeekstd_amd/cli.py)."""
import sys

from .cli import main

sys.exit(main())
