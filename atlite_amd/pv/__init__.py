"""Host-side pieces of the pv chain (orientation factories, solar-position tables)."""
