"""High-level flows over the hip backend (reference package: MuyGPyS/examples, deprecated there)."""
