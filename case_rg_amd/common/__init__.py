"""MI355X-native counterparts of the reference's ``common`` package (same module and class names)."""
