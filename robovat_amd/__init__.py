"""robovat_amd."""
