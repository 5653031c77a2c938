#!/usr/bin/env python3
"""Same command surface as the reference's sampling.py (task=generation|transcription|inpainting, dotted
key=value overrides); see diffroll_amd/cli.py."""
from diffroll_amd.cli import main

if __name__ == "__main__":
    main(default_task="generation")
