// Targets to add to WhisperKit's Package.swift (paths relative to this repository's root).
// .systemLibrary(name: "CWhisperHIP", path: "bindings/swift/Sources/CWhisperHIP"),
// .target(name: "WhisperKitHIP",
//         dependencies: ["WhisperKit", "CWhisperHIP"],
//         path: "bindings/swift/Sources/WhisperKitHIP",
//         linkerSettings: [.unsafeFlags(["-L", "whisperkit_amd", "-Xlinker", "-rpath", "-Xlinker", "whisperkit_amd"])]),
