"""`python -m frisk_amd ...` == the reference's `frisk ...` (frisk/__main__.py:16-18)."""
import sys

from .cli import main

if __name__ == "__main__":
    sys.exit(main())
