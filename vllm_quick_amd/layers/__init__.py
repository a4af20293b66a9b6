"""Host-side mirror of the reference's linear layers and AWQ plugin."""
