"""Host-side drivers of the SVI inner loop (one class per BASELINE config)."""
