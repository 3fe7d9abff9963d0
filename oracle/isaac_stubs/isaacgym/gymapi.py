"""Empty name-holder: the golden generator never creates a simulator."""
