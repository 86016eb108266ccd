"""Import-only stub so the reference module graph loads; never called on the ASR path."""
