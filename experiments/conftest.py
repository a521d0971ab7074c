def pytest_configure(config):
    config.addinivalue_line("markers", "experiments: parity tests of experiments/libpangu_experiments.so (needs a GPU)")
    config.addinivalue_line("markers", "gpu: needs a real MI355X")
