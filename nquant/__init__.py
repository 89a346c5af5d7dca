"""Namespace shim: the package lives in the directory `nquant.android_amd/` (the name the project layout asks for);
`import nquant.android_amd` resolves to it through nquant/android_amd/__init__.py."""
