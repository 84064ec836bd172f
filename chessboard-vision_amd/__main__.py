from .build import build

if __name__ == "__main__":
    print(build(verbose=True))
